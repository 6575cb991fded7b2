#!/usr/bin/env python3
"""Which value does every window column of a neighbour-frame tap wave of k_online4 (lws_online.hip) hold when it is used?

The engine is order-exact only if each column a tap wave multiplies is the value the reference's in-place sweep
(lwslib.cpp:1424-1492: sweeps in order, frames in order within a sweep, bins in order within a frame) would see there.  The
projection wave stores the two bins of bin pair u of (sweep s, frame rho) -- and the Hermitian images of bins 1..L and
N-L..N-1 in the pad columns -- at step t = DS s + SKS rho + u; a tap wave reads a two-step window of 16-byte cells (two
columns each) at three points of a pair of steps (it, it + 1):

    early   cells 0 .. NPRE-1        read at the end of the pair before: sees the stores of steps <= it - 3 for certain,
                                     those of step it - 2 perhaps (that step runs concurrently)
    even    cells NPRE .. NCELL-2    read before the sums of step it:   sees steps <= it - 2, perhaps it - 1
    odd     cell NCELL-1             read before the sums of step it+1: sees steps <= it - 1, perhaps it

plus the re-reads this script exists to justify (`extra`).  A column is right if the store it must see is certain and
the next store to the same place is not even "perhaps".  The model: every sweep covers every frame (the online driver's
sweeps cover a window of frames; what it leaves out only adds distance), rows other than the lane's own (the centre wave
and the projection wave deal with that one: they read it afresh every step).

usage: online_schedule_check.py            prints the table for every Q and lag the launcher can choose; exit code 1 on a miss
"""
import sys

L, SKS = 5, 4
WN = 2 * L + 2
NCELL = WN // 2 + 1


def reads_of(kind, odd_lag_fixes):
    """(cell, point) pairs in program order; point: 0 early, 1 even block, 2 odd block.  Mirrors tap_loop in k_online4."""
    npre = NCELL - 3 if kind == 1 else NCELL - 2
    r = [(c, 0) for c in range(npre)]
    r += [(c, 1) for c in range(npre, NCELL - 1)]
    if kind == 1:
        r.append((0, 1))                      # images of bins 4, 5 of frame rho-1 (an even start)
    if odd_lag_fixes and kind == 0:
        r.append((1, 1))                      # odd start: images of bins 4, 5 of frame rho+Q-1, previous sweep, at the minimal lag
    if kind == 1:
        r.append((NCELL - 2, 2))
    if odd_lag_fixes and kind == 1:
        r += [(1, 2), (2, 2)]                 # odd start: images of bins 2..5 of frame rho-1
    r.append((NCELL - 1, 2))
    return r


def check(Q, DS, NU=40, fixes=True, verbose=False):
    """Returns the list of misses (wave, start parity, u, column, what)."""
    N = 2 * (NU - 1)
    misses = []

    def sources(x):            # the bin whose store writes relative bin index x of a row (itself or the bin it is the image of)
        if 0 <= x <= N:
            return x
        if -L <= x < 0:
            return -x
        if N < x <= N + L:
            return 2 * N - x
        return None

    for r in range(1, Q):
        for h in (0, 1):
            kind = 1 if (r == 1 and h == 0) else 0      # frame rho-1: SKS steps ahead of the lane
            reads = reads_of(kind, fixes and (DS & 1))
            for s in (4, 5):                          # an even and (DS odd) an odd start
                rho = 10
                tstart = DS * s + SKS * rho
                rho2 = rho + r if h else rho - r
                s_req = s - 1 if h else s             # the sweep whose values this row must show
                par = tstart & 1
                for it in range(tstart - par, tstart + NU + 1, 2):
                    ue = it - tstart                  # bin pair at the even step of the pair
                    # horizon (steps certainly visible) of the read that supplies each cell at each of the two steps
                    hor_even, hor_odd = {}, {}
                    for cell, point in reads:
                        hz = (it - 3, it - 2, it - 1)[point]
                        if point <= 1:
                            hor_even[cell] = hz
                        hor_odd[cell] = hz
                    for step, hor, c0 in ((0, hor_even, 0), (1, hor_odd, 2)):
                        u = ue + step
                        if u < 0 or u >= NU:
                            continue
                        cols = list(range(c0, c0 + 2 * L + 2))
                        if kind == 1:
                            cols = cols[:-1]          # tap +L of the second bin: the projection wave adds it (it reads the column
                                                      # after its own stores of the step before: stored at step t-1 at the latest)
                        if u == NU - 1:
                            cols = cols[:2 * L + 1]   # (no second bin)
                        for col in cols:
                            x = 2 * ue + col - L      # relative bin index in the row
                            src = sources(x)
                            if src is None:
                                continue              # a pad column nobody writes (zero)
                            t_req = DS * s_req + SKS * rho2 + src // 2
                            t_next = t_req + DS
                            hz = hor[col // 2]
                            if t_req > hz:
                                misses.append(((r, h), par, u, col, "stale: stored at step %d, read sees <= %d" % (t_req, hz)))
                            elif t_next <= hz + 1:
                                misses.append(((r, h), par, u, col, "too new: next store at step %d, read sees <= %d (+1 perhaps)" % (t_next, hz)))
    return misses


def lags(Q):
    return range(SKS * Q + 1, SKS * Q + 18)   # from the launcher's minimum on (max(SKS Q + 1, what the sweep slots need))


def main():
    bad = 0
    for Q in range(2, 9):
        for DS in lags(Q):
            m_plain = check(Q, DS, fixes=False)
            m_fixed = check(Q, DS, fixes=True)
            tag = "ok" if not m_fixed else "MISS"
            print("Q=%d DS=%2d (%s): %s; without the odd-start re-reads: %d misses%s" % (
                Q, DS, "odd" if DS & 1 else "even", tag, len(m_plain),
                "" if not m_plain else "  e.g. wave (r=%d,h=%d) u=%d column %d: %s" % (*m_plain[0][0], m_plain[0][2], m_plain[0][3], m_plain[0][4])))
            if m_fixed or (not (DS & 1) and m_plain):
                bad += 1
                for m in m_fixed[:6]:
                    print("    ", m)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
