#!/usr/bin/env python3
"""The projection wave of the online engine's fourth layout (lws_online.hip, k_online4) enters its barrier with a COUNTED wait --
`s_waitcnt lgkmcnt(K); s_barrier` -- so that the LDS reads of the next step's operands stay in flight across the barrier while
the stores of this step are known to have landed.  That is only correct if the K youngest LDS / scalar-memory operations before
the wait are all reads (a wave's LDS operations complete in order).  The order is the compiler's to choose, so it is checked
in the compiled code, for every instantiation: `make -C lws_amd/csrc` runs this on the assembly of every build of the library
(a wrong order is a build failure, not a silent race), tests/test_online_isa.py runs it too.

usage: check_online_isa.py <lws_online.s>      exit code 0 = every counted wait is covered"""
import re
import sys

LGKM = re.compile(r"^\s*(ds_\w+|s_load_\w+|s_buffer_load_\w+|s_memtime|s_memrealtime|s_sendmsg\w*)\b")


def check(lines):
    """Returns (number of counted waits found, smallest margin = reads that follow the last store beyond the count)."""
    found, margin, worst = 0, None, 0
    counts = {}
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
        # the counts the source asks for (lws_online.hip: tap waves 5 / 3 reads of early cells, projection wave 7, 5 in the BIG variant);
        # any other count in front of a barrier is a wait this checker was not written for
        assert K in (3, 5, 7), "%s: unexpected `s_waitcnt lgkmcnt(%d); s_barrier` (line %d)" % (kernel, K, i + 1)
        counts[K] = counts.get(K, 0) + 1
        assert kernel and "k_online4" in kernel, (kernel, i)
        young = []
        j = i - 1
        while j >= 0:
            t = lines[j]
            if re.match(r"^\.LBB\S*:", t) or "s_cbranch" in t or "s_branch" in t:
                break
            mm = LGKM.match(t)
            if mm:
                if not mm.group(1).startswith("ds_read"):
                    break
                young.append(mm.group(1))
            j -= 1
        assert len(young) >= K, "%s: only %d LDS reads between the last store / join and `s_waitcnt lgkmcnt(%d); s_barrier` (line %d)" % (kernel, len(young), K, i + 1)
        assert K + 1 <= 15, "more than 15 operations in flight would wrap the 4-bit counter"
        # ... and everything issued since the last wait that bounded the counter is still countable: at most 15 in flight
        n, j, left = 0, i - 1, None
        while j >= 0:
            t = lines[j]
            mm = re.search(r"lgkmcnt\((\d+)\)", t)
            if "s_waitcnt" in t and mm:
                left = int(mm.group(1))
                break
            if LGKM.match(t):
                n += 1
            j -= 1
        assert left is not None and n + left <= 15, "%s: %d LDS / scalar-memory operations may be in flight at line %d (4-bit counter)" % (kernel, n + (left or 0), i + 1)
        worst = max(worst, n + left)
        margin = len(young) - K if margin is None else min(margin, len(young) - K)
    check.counts = counts
    return found, margin, worst


def main():
    lines = open(sys.argv[1]).read().split("\n")
    found, margin, worst = check(lines)
    assert found >= 6, "expected the counted waits of k_online4 (two half-steps per instantiation), found %d" % found
    assert all(check.counts.get(k, 0) > 0 for k in (3, 5, 7)), "a kind of counted wait is missing from the assembly: %r" % (check.counts,)
    print("check_online_isa: %d counted waits %r, each followed by its stores' completion; smallest margin %d reads; at most %d operations in flight"
          % (found, check.counts, margin, worst))


if __name__ == "__main__":
    try:
        main()
    except AssertionError as e:
        print("check_online_isa: FAILED: %s" % (e,), file=sys.stderr)
        sys.exit(1)
