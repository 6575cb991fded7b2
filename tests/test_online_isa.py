"""CPU (needs hipcc, no GPU): the counted wait before the barrier of the online engine's projection wave (`s_waitcnt lgkmcnt(K);
s_barrier`, lws_online.hip) is checked in the COMPILED code of every instantiation by tools/check_online_isa.py -- the Makefile
runs the same check on every build of the library; here it is run on a fresh compilation, and on doctored assembly to show
that it does catch a store behind the wait."""
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lws_amd", "csrc", "lws_online.hip")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_online_isa  # noqa: E402


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_counted_wait_before_the_barrier_covers_the_stores():
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-I", os.path.join(ROOT, "include"),
               "--cuda-device-only", "-S", SRC, "-o", os.path.join(td, "online.s")]
        subprocess.run(cmd, check=True, cwd=td, capture_output=True)
        lines = open(os.path.join(td, "online.s")).read().split("\n")
    found, margin, worst = check_online_isa.check(lines)
    assert found >= 8, found      # two half-steps per instantiation: Q in {2, 4, 8} and the BIG variants
    assert margin >= 0, margin    # (the tap waves' waits count exactly the reads that follow their store -- pinned in the source by a
                                  #  compiler barrier between the two; the projection wave's BIG variant keeps a margin of three: lgkmcnt(5))
    assert worst <= 15, worst     # operations in flight at a counted wait: the counter has four bits


def test_checker_catches_a_store_behind_the_wait():
    good = ["_ZN3lws9k_online4ILi4ELi5ELb0ELb0EEEvNS_10OnlineArgsE:", " s_waitcnt lgkmcnt(0)", " ds_write_b64 v1, v[2:3]"] + \
           [" ds_read_b128 v[4:7], v1"] * 7 + [" s_waitcnt lgkmcnt(7)", " s_barrier"]
    assert check_online_isa.check(good) == (1, 0, 8)
    bad = list(good)
    bad.insert(6, " ds_write_b64 v1, v[2:3]")      # a store among the seven youngest operations
    with pytest.raises(AssertionError):
        check_online_isa.check(bad)
    crowded = good[:3] + [" ds_read_b128 v[4:7], v1"] * 9 + good[3:]      # 17 operations since the last full wait
    with pytest.raises(AssertionError):
        check_online_isa.check(crowded)
