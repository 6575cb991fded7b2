"""CPU (needs hipcc, no GPU): the projection wave of the online engine's fourth layout (lws_online.hip, k_online4) enters its
barrier with a COUNTED wait -- `s_waitcnt lgkmcnt(K); s_barrier` -- so that the LDS reads of the next step's operands stay in
flight across the barrier while the stores of this step are known to have landed.  That is only correct if the K youngest
LDS / scalar-memory operations before the wait are all reads (a wave's LDS operations complete in order): checked here in the
compiled code, for every instantiation, because the order is the compiler's to choose."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "lws_amd", "csrc", "lws_online.hip")
LGKM = re.compile(r"^\s*(ds_\w+|s_load_\w+|s_buffer_load_\w+|s_memtime|s_memrealtime|s_sendmsg\w*)\b")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_counted_wait_before_the_barrier_covers_the_stores():
    with tempfile.TemporaryDirectory() as td:
        cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
               "--cuda-device-only", "-S", SRC, "-o", os.path.join(td, "online.s")]
        subprocess.run(cmd, check=True, cwd=td, capture_output=True)
        lines = open(os.path.join(td, "online.s")).read().split("\n")
    found = 0
    kernel = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_ZN3lws\S*k_online\S*):", ln)
        if m:
            kernel = m.group(1)
        m = re.match(r"^\s*s_waitcnt lgkmcnt\((\d+)\)\s*$", ln)
        if not m or int(m.group(1)) == 0:
            continue
        nxt = next((l for l in lines[i + 1:i + 4] if l.strip() and not l.strip().startswith(";")), "")
        if "s_barrier" not in nxt:
            continue            # (the compiler's own partial waits)
        K = int(m.group(1))
        found += 1
        assert kernel and "k_online4" in kernel, (kernel, i)
        young = []
        j = i - 1
        while j >= 0 and len(young) < K:
            t = lines[j]
            assert not re.match(r"^\.LBB\S*:", t) and "s_cbranch" not in t and "s_branch" not in t, \
                "%s: fewer than %d LDS operations between the last join and the counted wait (line %d)" % (kernel, K, i)
            mm = LGKM.match(t)
            if mm:
                young.append(mm.group(1))
            j -= 1
        assert len(young) == K and all(op.startswith("ds_read") for op in young), (kernel, K, young)
    assert found >= 6, found      # two half-steps per instantiation, Q in {2, 4, 8}
