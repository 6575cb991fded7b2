"""CPU: the host-side check that decides which engine a plan gets (lws_weights_structure -> lws::weights_twiddle): create_weights'
tensors -- summarised, or general with one row per bin (lws.pyx:160-181) -- have W[p][r][k] == W[0][r][k] exp(2j pi p r step / period)
with period / step = frame / hop in lowest terms; anything else has no structure."""
import math

import numpy as np
import pytest

import lws_amd
from lws_amd import _capi


@pytest.mark.parametrize("fsize,fshift", [(64, 16), (64, 32), (64, 8), (48, 16), (1024, 256), (400, 160), (512, 160), (1024, 384), (1000, 400),
                                          (1000, 200), (768, 128), (1024, 160), (512, 300), (1024, 176), (2048, 768), (100, 30), (1024, 100)])
def test_create_weights_tensors_have_the_structure(fsize, fshift):
    p = lws_amd.lws(fsize, fshift)
    g = math.gcd(fsize, fshift)
    for W in (p.W, p.W_af):
        got = _capi.weights_structure(W)
        assert got == (fsize // g, (fshift // g) % (fsize // g)), (got, fsize, fshift)
    ai = _capi.weights_structure(p.W_ai)          # (hop >= half the frame: no neighbour-frame weights at all -> fits any twiddle)
    assert ai in ((fsize // g, (fshift // g) % (fsize // g)), (0, 0)), ai
    if fsize % fshift == 0:                        # general tensors of an integer Q are periodic copies of the summarised ones
        pg = lws_amd.lws(fsize, fshift, use_simplifications=False)
        assert pg.W.shape[0] == fsize and _capi.weights_structure(pg.W) == (fsize // fshift, 1)


def test_tensors_without_the_structure_are_recognised():
    p = lws_amd.lws(64, 16)
    W = np.array(p.W)
    W[1, 1, 3] *= 1.01
    assert _capi.weights_structure(W) is None
    W = np.array(lws_amd.lws(400, 160).W)
    W[37, 1, 0] += 1e-6
    assert _capi.weights_structure(W) is None
    rng = np.random.default_rng(0)
    assert _capi.weights_structure(rng.standard_normal((4, 4, 6)) + 1j * rng.standard_normal((4, 4, 6))) is None
    # a structured tensor built by hand: any base weights, period 7, step 3
    base = rng.standard_normal((3, 6)) + 1j * rng.standard_normal((3, 6))
    pp = np.arange(14)[:, None, None]
    Wt = base[None] * np.exp(2j * np.pi * pp * np.arange(3)[None, :, None] * 3 / 7)
    assert _capi.weights_structure(Wt) == (7, 3)
