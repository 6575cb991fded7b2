"""Schedule model of the fp64 systolic batch engine (lws_amd/csrc/lws_sys64.hip), lane for lane, in numpy.

Development tool (runs on the CPU, uses the oracle as the checker): the kernel is a transcription of `run_pass` below,
so a change of the schedule is tried here first.  What is modelled: one wave per sweep slot, lane = frame (64 frames in
flight, 8 steps apart), one bin per step; the taps of the neighbour frames arrive in *scatter* form (the position a step
consumes is added to the 2L+1 bins it reaches), the taps of the frame itself in gather form from two register windows.
The model keeps every slot's output by frame-time row and asserts the ages the LDS rings of the kernel must hold.

    python tools/sys64_model.py            # a few shapes against the oracle
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = 5
NL = 64      # lanes = frames in flight per sweep slot
SK = 8       # steps between consecutive frames
MARG = 64


def geometry(F, Q):
    P = max(NL * SK, -(-(F + L) // 8) * 8)      # a frame's last images are written during the first steps of the lane's next frame
    gap = P - NL * SK
    LAG = -(-(L + SK * (Q - 1) + gap + 2) // 8) * 8
    R = LAG - L + 1
    return P, gap, LAG, R


def to_skew(ext, F, Q):
    Tp = ext.shape[0]
    P, gap, LAG, R = geometry(F, Q)
    nblk = -(-Tp // NL)
    U = SK * (NL - 1) + P * nblk + 8
    G = np.zeros((U + 2 * MARG, NL), complex)
    for me in range(Tp):
        j, blk = me % NL, me // NL
        base = SK * j + P * blk + L + MARG
        G[base:base + F + L, j] = ext[me]
    return G, U, nblk


def from_skew(G, Tp, F, Q):
    P = geometry(F, Q)[0]
    out = np.empty((Tp, F), complex)
    for me in range(Tp):
        j, blk = me % NL, me // NL
        base = SK * j + P * blk + L + MARG
        out[me] = G[base:base + F, j]
    return out


def extend_cols(S, Q):
    """frames clamped at both ends (lwslib.cpp:15-40); columns: bins 0..F-1 and the L images above Nyquist"""
    T, F = S.shape
    rows = np.clip(np.arange(T + 2 * (Q - 1)) - (Q - 1), 0, T - 1)
    ext = np.empty((len(rows), F + L), complex)
    ext[:, :F] = S[rows]
    for j in range(1, L + 1):
        ext[:, F - 1 + j] = np.conj(ext[:, F - 1 - j])
    return ext


def pair_sd(ax, ay, wx, wy, sx, dy, sy, dx):
    return ax + (wx * sx - wy * dy), ay + (wx * sy + wy * dx)


def run_pass(G, A, Wm, thr, F, T, Q, ns, stats):
    """ns sweeps (thr[0..ns-1]) over the skewed state G, in place."""
    Qp = Wm.shape[0]
    Tp = T + 2 * (Q - 1)
    P, gap, LAG, R = geometry(F, Q)
    nblk = -(-Tp // NL)
    U = SK * (NL - 1) + P * nblk + 8      # (+ 8: the last frame's images)
    lane = np.arange(NL)
    X = [np.zeros((U + 2 * MARG, NL), complex) for _ in range(ns)]        # slot outputs by frame-time row (LDS rings start as zeros)
    Xtime = [np.full(U + 2 * MARG, -10 ** 9) for _ in range(ns)]         # time a row was written (ring-age checks)
    acc = [np.zeros((2 * L + 1, NL), complex) for _ in range(ns)]
    cn = [np.zeros((L + 1, NL), complex) for _ in range(ns)]
    co = [np.zeros((L + 1, NL), complex) for _ in range(ns)]

    def rd(arr, rows, lanes):
        return arr[rows + MARG, lanes]

    for t in range(U + LAG * (ns - 1)):
        for s in range(ns):
            u = t - LAG * s
            if u < 0 or u >= U:
                continue
            v = u - SK * lane
            act = (v >= 0) & (v < P * nblk)
            blk = np.where(v >= 0, v // P, 0)
            w = np.where(v >= 0, v % P, 0)
            me = NL * blk + lane
            c = w - L
            ph = u % 8
            assert np.all((w[act] % 8) == ph)
            pos_ok = act & (w <= F + L - 1)
            for d in range(2 * L + 1):              # a lane starts its next frame with empty sums
                acc[s][d] = np.where(act & (w == 0), 0, acc[s][d])
            prev = G if s == 0 else X[s - 1]

            def age_prev(rows, sel):
                if s == 0 or not np.any(sel):
                    return
                a = t - Xtime[s - 1][rows[sel] + MARG]
                stats['prev_min'] = min(stats['prev_min'], a.min()); stats['prev_max'] = max(stats['prev_max'], a.max())

            def age_own(rows, sel):
                if not np.any(sel):
                    return
                a = t - Xtime[s][rows[sel] + MARG]
                stats['own_min'] = min(stats['own_min'], a.min()); stats['own_max'] = max(stats['own_max'], a.max())

            useful = pos_ok & (me >= Q - 1) & (me < T + Q - 1)
            # (a) the frame's own old value at position w, with the images above Nyquist that this sweep already rewrote
            rows = u + L + 0 * lane
            O = rd(prev, rows, lane)
            age_prev(rows, useful)
            kk = 2 * c + L - 2 * (F - 1)          # steps since the source of the image at c + L was updated
            newimg = (kk >= 1) & (kk <= L)
            for k in range(1, L + 1):
                O = np.where(newimg & (kk == k), np.conj(cn[s][k]), O)
            co[s][L] = O
            # (b) neighbour frames: position w of frames me -+ r, scattered to bins c .. c + 2L
            for r in range(1, Q):
                rowsL = u + L - SK * r - np.where(lane < r, gap, 0)
                rowsR = u + L + SK * r + np.where(lane + r >= NL, gap, 0)
                Lv = rd(X[s], rowsL, (lane - r) % NL)
                Rv = rd(prev, rowsR, (lane + r) % NL)
                age_own(rowsL, useful); age_prev(rowsR, useful)
                # (no masking: rows that are no position of a frame hold zeros)
                sx = Lv.real + Rv.real; dy = Lv.imag - Rv.imag
                sy = Lv.imag + Rv.imag; dx = Lv.real - Rv.real
                for d in range(2 * L + 1):
                    tgt = (ph - L + d) % Qp
                    if d < L:
                        wv = np.conj(Wm[(Qp - tgt) % Qp, r, L - d])
                    else:
                        wv = Wm[tgt, r, d - L]
                    ax, ay = pair_sd(acc[s][d].real, acc[s][d].imag, wv.real, wv.imag, sx, dy, sy, dx)
                    acc[s][d] = ax + 1j * ay
                # images below DC: position -w is the conjugate of position w, reaches bins 0 .. L - w
                if 1 <= ph <= L:
                    z = act & (w == ph)
                    for ct in range(0, L - ph + 1):
                        d = ct + L - ph
                        wv = Wm[ct % Qp, r, ct + ph]
                        ax = acc[s][d].real + (wv.real * sx + wv.imag * dy)
                        ay = acc[s][d].imag + (-wv.real * sy + wv.imag * dx)
                        acc[s][d] = np.where(z, ax + 1j * ay, acc[s][d])
            # (c) the frame itself
            a = acc[s][0].copy()
            rowc = (ph - L) % Qp
            for k in range(1, L + 1):
                b = cn[s][k].copy()
                for cc in range(0, L):            # bins 0..L-1: images below DC
                    if k > cc:
                        q = k - cc
                        src = np.conj(cn[s][cc - q]) if q < cc else np.conj(co[s][q - cc])
                        b = np.where(c == cc, src, b)
                wv = Wm[rowc, 0, k]
                cv = co[s][k]
                ax, ay = pair_sd(a.real, a.imag, wv.real, wv.imag, b.real + cv.real, b.imag - cv.imag,
                                 b.imag + cv.imag, b.real - cv.real)
                a = ax + 1j * ay
            amp = rd(A, u + 0 * lane, lane)
            mag = np.sqrt(a.real * a.real + a.imag * a.imag)
            upd = act & (c >= 0) & (c <= F - 1) & (me >= Q - 1) & (me < T + Q - 1) & (amp > thr[s]) & (mag > 0)
            with np.errstate(all='ignore'):
                vnew = (a.real * amp / mag) + 1j * (a.imag * amp / mag)
            val = np.where(upd, vnew, co[s][0])
            # images above Nyquist are written when the lane passes them
            jj = np.where(c < 0, c + P, c) - (F - 1)      # (c < 0: still the images of the frame the lane has just left)
            img = (v >= 0) & np.where(c < 0, me >= NL, act) & (jj >= 1) & (jj <= L)
            rowsI = u - 2 * np.where(img, jj, 0)
            age_own(rowsI, img & (me >= Q - 1) & (me < T + Q - 1))
            val = np.where(img, np.conj(rd(X[s], rowsI, lane)), val)
            # windows
            j2 = (F - 1) - c
            for q in (1, 2):
                if 2 * q <= L:
                    co[s][2 * q] = np.where(act & (j2 == q), np.conj(val), co[s][2 * q])
            wr = (act & (c >= 0) & (c <= F + L - 1)) | img
            X[s][u + MARG] = np.where(wr, val, 0)
            Xtime[s][u + MARG] = t
            if s == ns - 1:
                G[u + MARG, wr] = val[wr]
            for k in range(L, 1, -1):
                cn[s][k] = cn[s][k - 1]
            cn[s][1] = val
            for k in range(0, L):
                co[s][k] = co[s][k + 1]
            for d in range(2 * L):
                acc[s][d] = acc[s][d + 1]
            acc[s][2 * L] = 0


def run_continuous(G, A, Wm, thr, F, T, Q, NS, stats):
    """ALL sweeps in one go: slot s takes sweeps s, s + NS, s + 2 NS, .. one after the other without draining -- a lane starts frame
    j of its next sweep the step after it finished its last frame, so the 8 x 63 steps in which the lanes of a pass start and stop
    one after the other are paid once, not once per NS sweeps.  Slot 0 reads what slot NS - 1 wrote to the skewed state during
    the sweep before (P x blocks - LAG (NS - 1) steps earlier); rings are addressed by global time as before."""
    n = len(thr)
    Qp = Wm.shape[0]
    Tp = T + 2 * (Q - 1)
    P, gap, LAG, R = geometry(F, Q)
    nblk = -(-Tp // NL)
    PS = P * nblk                                   # steps a lane spends on a sweep
    lane = np.arange(NL)
    Kmax = -(-n // NS)
    t_end = LAG * (NS - 1) + SK * (NL - 1) + Kmax * PS + 8
    assert PS > SK * (Q - 1) + gap + L + LAG * (NS - 1) + 8
    X = [np.zeros((t_end + 2 * MARG, NL), complex) for _ in range(NS)]      # slot outputs by GLOBAL time
    acc = [np.zeros((2 * L + 1, NL), complex) for _ in range(NS)]
    cn = [np.zeros((L + 1, NL), complex) for _ in range(NS)]
    co = [np.zeros((L + 1, NL), complex) for _ in range(NS)]
    thr = np.asarray(thr, float)

    def rdt(arr, times, lanes):                     # ring read by global time (negative times: the ring's initial zeros)
        return arr[times + MARG, lanes]

    for t in range(t_end):
        for s in range(NS):
            Ks = len(range(s, n, NS))               # sweeps of this slot
            v = t - LAG * s - SK * lane             # lane time
            started = v >= 0
            k = np.where(started, v // PS, 0)
            vv = np.where(started, v % PS, 0)
            blk = vv // P
            w = vv % P
            act = started & (k < Ks)
            me = NL * blk + lane
            c = w - L
            g = s + k * NS                          # the sweep a lane is in
            thr_l = thr[np.minimum(g, n - 1)]
            ph = (t - LAG * s) % 8
            assert np.all((w[started] % 8) == ph)
            for d in range(2 * L + 1):
                acc[s][d] = np.where(started & (w == 0), 0, acc[s][d])
            row = SK * lane + P * blk + w           # row of (me, c) in the skewed state, minus L
            # (a) own old value at position w
            if s == 0:
                O = G[row + L + MARG, lane]
            else:
                O = rdt(X[s - 1], t - (LAG - L) + 0 * lane, lane)
                stats['prev_max'] = max(stats['prev_max'], LAG - L)
            kk = 2 * c + L - 2 * (F - 1)
            newimg = (kk >= 1) & (kk <= L)
            for q in range(1, L + 1):
                O = np.where(newimg & (kk == q), np.conj(cn[s][q]), O)
            co[s][L] = O
            # (b) neighbours
            for r in range(1, Q):
                wrapL = np.where(lane < r, gap, 0)
                wrapR = np.where(lane + r >= NL, gap, 0)
                Lv = rdt(X[s], t - (SK * r - L) - wrapL, (lane - r) % NL)
                if s == 0:
                    Rv = G[row + L + SK * r + wrapR + MARG, (lane + r) % NL]
                else:
                    Rv = rdt(X[s - 1], t - (LAG - L - SK * r) + wrapR, (lane + r) % NL)
                    stats['prev_min'] = min(stats['prev_min'], LAG - L - SK * r - gap)
                sx = Lv.real + Rv.real; dy = Lv.imag - Rv.imag
                sy = Lv.imag + Rv.imag; dx = Lv.real - Rv.real
                for d in range(2 * L + 1):
                    tgt = (ph - L + d) % Qp
                    wv = np.conj(Wm[(Qp - tgt) % Qp, r, L - d]) if d < L else Wm[tgt, r, d - L]
                    ax, ay = pair_sd(acc[s][d].real, acc[s][d].imag, wv.real, wv.imag, sx, dy, sy, dx)
                    acc[s][d] = ax + 1j * ay
                if 1 <= ph <= L:
                    z = started & (w == ph)
                    for ct in range(0, L - ph + 1):
                        d = ct + L - ph
                        wv = Wm[ct % Qp, r, ct + ph]
                        ax = acc[s][d].real + (wv.real * sx + wv.imag * dy)
                        ay = acc[s][d].imag + (-wv.real * sy + wv.imag * dx)
                        acc[s][d] = np.where(z, ax + 1j * ay, acc[s][d])
            # (c) the frame itself
            a = acc[s][0].copy()
            rowc = (ph - L) % Qp
            for q in range(1, L + 1):
                b = cn[s][q].copy()
                for cc in range(0, L):
                    if q > cc:
                        qq = q - cc
                        src = np.conj(cn[s][cc - qq]) if qq < cc else np.conj(co[s][qq - cc])
                        b = np.where(c == cc, src, b)
                wv = Wm[rowc, 0, q]
                cv = co[s][q]
                ax, ay = pair_sd(a.real, a.imag, wv.real, wv.imag, b.real + cv.real, b.imag - cv.imag, b.imag + cv.imag, b.real - cv.real)
                a = ax + 1j * ay
            amp = A[row + MARG, lane]
            mag = np.sqrt(a.real * a.real + a.imag * a.imag)
            upd = act & (c >= 0) & (c <= F - 1) & (me >= Q - 1) & (me < T + Q - 1) & (amp > thr_l) & (mag > 0)
            with np.errstate(all='ignore'):
                vnew = (a.real * amp / mag) + 1j * (a.imag * amp / mag)
            val = np.where(upd, vnew, co[s][0])
            # images above Nyquist: of the current frame, or -- during the first steps of a frame -- of the frame the lane has just
            # left (the last frame of its previous sweep when this is the first block)
            before = c < 0
            had_prev = (blk > 0) | (k > 0)
            jj = np.where(before, c + P, c) - (F - 1)
            img = started & np.where(before, had_prev & ((k < Ks) | ((k == Ks) & (blk == 0))), act) & (jj >= 1) & (jj <= L)
            val = np.where(img, np.conj(rdt(X[s], t - 2 * np.where(img, jj, 0), lane)), val)
            j2 = (F - 1) - c
            for q in (1, 2):
                if 2 * q <= L:
                    co[s][2 * q] = np.where(act & (j2 == q), np.conj(val), co[s][2 * q])
            wr = (act & (c >= 0) & (c <= F + L - 1)) | img
            X[s][t + MARG] = np.where(wr, val, 0)
            # the state in HBM: written by the last slot of a pass, and by whoever runs the very last sweep
            g_w = np.where(before, g - NS, g)       # an image of the previous frame belongs to the sweep before at a sweep change
            g_w = np.where(before & (blk > 0), g, g_w)
            last = (s == NS - 1) | (g_w == n - 1)
            roww = np.where(before & (blk == 0), SK * lane + P * nblk + w, row)
            sel = wr & last
            G[roww[sel] + MARG, lane[sel]] = val[sel]
            for q in range(L, 1, -1):
                cn[s][q] = cn[s][q - 1]
            cn[s][1] = val
            for q in range(0, L):
                co[s][q] = co[s][q + 1]
            for d in range(2 * L):
                acc[s][d] = acc[s][d + 1]
            acc[s][2 * L] = 0


def batch_lws_model(S, W, thresholds, NS=3, continuous=False):
    S = np.asarray(S, complex)
    T, F = S.shape
    Qp, Q, _ = W.shape
    Wm = np.where(np.abs(W) > 1e-12, W, 0)
    ext = extend_cols(S, Q)
    G, U, nblk = to_skew(ext, F, Q)
    A, _, _ = to_skew(np.abs(ext).astype(complex), F, Q)
    A = A.real.copy()
    mean = float(np.mean(np.abs(S)))
    thr = [th * mean for th in thresholds]
    stats = dict(prev_min=10 ** 9, prev_max=-1, own_min=10 ** 9, own_max=-1)
    if continuous:
        run_continuous(G, A, Wm, thr, F, T, Q, NS, stats)
    else:
        for i in range(0, len(thr), NS):
            run_pass(G, A, Wm, thr[i:i + NS], F, T, Q, len(thr[i:i + NS]), stats)
    out = from_skew(G, T + 2 * (Q - 1), F, Q)[Q - 1:Q - 1 + T]
    # the images above Nyquist in the skewed state are those of the final values (they go back into the extended buffers)
    P = geometry(F, Q)[0]
    for me in range(Q - 1, T + Q - 1):
        j, blk = me % NL, me // NL
        base = SK * j + P * blk + L + MARG
        assert np.array_equal(G[base + F:base + F + L, j], np.conj(G[base + F - 2:base + F - 2 - L:-1, j])), me
    return out, stats


if __name__ == "__main__":
    import lws_amd
    from oracle.oracle import Oracle
    orc = Oracle()
    rng = np.random.default_rng(1)
    for (fs, hop, T, F, it) in [(64, 16, 9, 33, 4), (64, 16, 70, 33, 5), (64, 32, 67, 33, 4), (1024, 256, 12, 513, 4), (1024, 256, 70, 513, 4), (1012, 253, 66, 507, 3), (1020, 255, 66, 511, 3)]:
        p = lws_amd.lws(fs, hop, batch_iterations=it, batch_alpha=1.0)
        S = rng.standard_normal((T, F)) + 1j * rng.standard_normal((T, F))
        thr = lws_amd.get_thresholds(it, 1.0, 0.1, 1)
        ref = orc.batch_lws(S, p.W, thr)
        out, st = batch_lws_model(S, p.W, thr)
        err = np.abs(out - ref).max() / np.abs(ref).max()
        outc, _ = batch_lws_model(S, p.W, thr, NS=3, continuous=True)
        print("   continuous: max rel err %.2e" % (np.abs(outc - ref).max() / np.abs(ref).max()))
        print("lws(%d,%d) T=%d F=%d iters=%d Q=%d: max rel err %.2e  ages %s  geometry(P,gap,LAG,R)=%s"
              % (fs, hop, T, F, it, p.W.shape[1], err, st, geometry(F, p.W.shape[1])))
