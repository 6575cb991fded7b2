"""The schedule model of the online engine's tap waves (tools/online_schedule_check.py): for every Q and every lag between
sweeps the launcher can choose, each window column a neighbour-frame tap wave multiplies must be the value the reference's
in-place sweep (lwslib.cpp:1424-1492) sees there -- stored for certain before the read, and not overwritten by a store the
read may or may not see.  CPU only: it is arithmetic on step numbers."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("online_schedule_check", os.path.join(ROOT, "tools", "online_schedule_check.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("Q", range(2, 9))
def test_every_column_is_the_reference_orders_value(Q):
    m = _model()
    for DS in m.lags(Q):
        assert m.check(Q, DS, fixes=True) == [], (Q, DS)
        for NU in (9, 17, 33):       # short frames: the two frame edges are close to each other
            assert m.check(Q, DS, NU=NU, fixes=True) == [], (Q, DS, NU)


@pytest.mark.parametrize("Q", range(2, 9))
def test_even_lags_need_no_re_reads_and_odd_lags_do(Q):
    m = _model()
    for DS in m.lags(Q):
        misses = m.check(Q, DS, fixes=False)
        if DS % 2 == 0:
            assert misses == []
        else:
            # the images of bins 2..5 (window columns 2..5 of a lane that starts at an odd step) of the two frames that are close
            # ahead: frame rho-1 at every odd lag, frame rho+Q-1 of the previous sweep at the smallest odd lag
            waves = {w for (w, _par, _u, _col, _what) in misses}
            assert (1, 0) in waves and waves <= {(1, 0), (Q - 1, 1)}
            assert all(par == 1 and u == 0 and 2 <= col <= 5 for (_w, par, u, col, _what) in misses)
            if DS > 4 * Q + 1:
                assert waves == {(1, 0)}


def test_a_lag_below_the_launchers_minimum_is_caught():
    """SKS Q steps are order-exact, but only if frame rho+Q-1's wave is treated as frame rho-1's is (built in round 4: slower)."""
    m = _model()
    for Q in range(2, 9):
        assert {w for (w, *_rest) in m.check(Q, 4 * Q, fixes=True)} == {(Q - 1, 1)}, Q


def test_the_model_reads_what_the_kernel_reads():
    """The re-reads the model calls `fixes` are in tap_loop of k_online4, in the odd-lag build, for the pair in which a lane starts at the odd step."""
    src = open(os.path.join(ROOT, "lws_amd", "csrc", "lws_online.hip")).read()
    assert "if constexpr (ODD && KIND == 0) { if (ue == -1) ld(w, std::integral_constant<int, 1>{}); }" in src
    assert ("if constexpr (ODD && KIND == 1) { if (ue == -1) { ld(w, std::integral_constant<int, 1>{}); "
            "ld(w, std::integral_constant<int, 2>{}); } }") in src
    assert "NPRE = KIND == 1 ? NCELL - 3 : NCELL - 2" in src
    m = _model()
    assert m.NCELL == 7 and m.SKS == 4 and m.L == 5
