"""The online engine (k_online4) with the smallest order-exact lag between sweeps against the same engine with one, two and
three steps more (LWS_ONLINE_LAG_PLUS: even and odd lags, and lags at which no frame is "just ahead"): the schedules compute the
same sums in the same order (TF_RTISI_LA, lwslib.cpp:1424-1492), so the results must agree bit for bit -- any stale or too-new
window column shows here."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [  # (fsize, fshift): Q = 4 and Q = 2 with static twiddles, then the table-twiddle variant (Q = 3, 5, 6, fractional Q)
    (512, 128), (1024, 256), (2048, 512), (64, 16), (256, 128), (1024, 512),
    (768, 256), (1000, 200), (960, 160), (1024, 384), (400, 160), (896, 128), (960, 128),
    (4096, 1024),       # the BIG variant (magnitudes and step table not in LDS)
]


def _online(fsize, fshift, S, LA, nit):
    import lws_amd
    os.environ["LWS_ONLINE_LAYOUT"] = "4"       # (the launcher weighs the layouts by their lags: keep it from changing engines)
    p = lws_amd.lws(fsize, fshift, mode="music", online_iterations=nit, look_ahead=LA)
    try:
        out = np.asarray(p.online_lws(S))
    finally:
        os.environ.pop("LWS_ONLINE_LAYOUT", None)
    return out, p.plan().last_kernel()["name"]


@pytest.mark.parametrize("fsize,fshift", SHAPES)
def test_odd_lag_equals_even_lag_bit_for_bit(fsize, fshift):
    rng = np.random.default_rng(fsize * 7 + fshift)
    F = fsize // 2 + 1
    for case in range(3 if fsize > 2048 else 6):
        T = int(rng.integers(1, 48)); B = int(rng.integers(1, 4)); LA = int(rng.integers(0, 6)); nit = int(rng.integers(1, 10))
        S = rng.rayleigh(1.0, (B, T, F)).astype(np.complex128)
        os.environ.pop("LWS_ONLINE_LAG_PLUS", None)
        a, name = _online(fsize, fshift, S, LA, nit)
        assert name.startswith("online_lds"), name
        assert np.isfinite(a).all()
        for plus in (1, 2, 3):
            os.environ["LWS_ONLINE_LAG_PLUS"] = str(plus)
            try:
                b, _ = _online(fsize, fshift, S, LA, nit)
            finally:
                os.environ.pop("LWS_ONLINE_LAG_PLUS", None)
            assert np.array_equal(a, b), (fsize, fshift, T, B, LA, nit, plus, float(np.abs(a - b).max()))
