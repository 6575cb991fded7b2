#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE implementation.

Run in the build container only (needs /root/reference, gcc, Cython):

    python tests/golden/make_golden.py

It (1) builds the reference's Python extension in a temporary directory exactly as its own
setup.py does (nothing is copied into this repository), (2) compiles the
reference's lwslib.cpp into oracle/_ref/liblws_ref.so via oracle/Makefile for direct kernel
access, and (3) writes small .npz fixtures: inputs, weights, thresholds and the reference's fp64
outputs.  The fixtures are data only; they are what travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("LWS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def build_reference_module():
    tmp = tempfile.mkdtemp(prefix="lws_ref_build_")
    for f in ("lws.pyx", "lwslib.pxd", "setup.py", "README.md", "MANIFEST.in"):
        shutil.copy(os.path.join(REF, "python", f), tmp)
    shutil.copytree(os.path.join(REF, "lwslib"), os.path.join(tmp, "lwslib"))
    subprocess.run(["chmod", "-R", "u+w", tmp], check=True)
    env = dict(os.environ, LWS_USE_CYTHON="1")
    subprocess.run([sys.executable, "setup.py", "build_ext", "-i"], cwd=tmp, env=env, check=True,
                   capture_output=True)
    sys.path.insert(0, tmp)
    import lws as ref  # noqa
    assert ref.__version__ == "1.2.8"
    return ref


def cplx(rng, shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


def main():
    ref = build_reference_module()
    from oracle.oracle import RefLib, build as build_oracle
    build_oracle()
    rl = RefLib()
    rng = np.random.default_rng(20260928)

    # ------------------------------------------------------------------ helpers
    out = {}
    cfgs = [(64, 16), (64, 32), (64, 8), (48, 16), (64, 24)]  # Q = 4, 2, 8, 3, fractional (2.67)
    for fsize, fshift in cfgs:
        p = ref.lws(fsize, fshift)
        key = f"{fsize}_{fshift}"
        out[f"awin_{key}"] = p.awin
        out[f"swin_{key}"] = p.swin
        out[f"W_{key}"] = p.W
        out[f"W_ai_{key}"] = p.W_ai
        out[f"W_af_{key}"] = p.W_af
        out[f"win_ai_{key}"] = p.win_ai
        out[f"win_af_{key}"] = p.win_af
        out[f"Wgen_{key}"] = ref.create_weights(p.awin, p.swin, fshift, 3, use_summarized_weights=False)
    out["hann_sym_16"] = ref.hann(16)
    out["hann_asym_16"] = ref.hann(16, symmetric=False)
    out["hann_asym_off_16"] = ref.hann(16, symmetric=False, use_offset=True)
    out["thr_100"] = ref.get_thresholds(100, 100, 0.1, 1)
    out["thr_gamma"] = ref.get_thresholds(7, 2.0, 0.3, 1.5)
    x = rng.standard_normal(700)
    out["x"] = x
    p = ref.lws(64, 16)
    X = p.stft(x)
    out["stft_64_16"] = X
    out["istft_64_16"] = p.istft(X)
    out["stft_np_64_16"] = ref.stft(x, 64, 16, p.awin, perfectrec=False)
    out["istft_np_64_16"] = ref.istft(out["stft_np_64_16"], 16, p.swin, perfectrec=False)
    out["consistency_64_16"] = np.array(p.get_consistency(np.abs(X).astype(complex)))
    S = cplx(rng, (6, 9))
    out["ext_in"] = S
    out["ext_L2_Q3"] = ref.extspec(S, 2, 3)
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)

    # ------------------------------------------------------------------ single sweeps (kernel level)
    out = {}
    cases = []
    for (fsize, fshift) in [(64, 32), (64, 16), (64, 8)]:
        p = ref.lws(fsize, fshift, L=5)
        Q = fsize // fshift
        for (T, F) in ([(10, 33), (14, 33)] if fshift == 16 else [(10, 33)]):
            S = cplx(rng, (T, F))
            cases.append((fsize, fshift, Q, T, F, 5, p, S))
    p1 = ref.lws(32, 8, L=1)
    cases.append((32, 8, 4, 12, 17, 1, p1, cplx(rng, (12, 17))))
    for ci, (fsize, fshift, Q, T, F, L, p, S) in enumerate(cases):
        tag = f"c{ci}"
        out[f"{tag}_meta"] = np.array([fsize, fshift, Q, T, F, L])
        out[f"{tag}_S"] = S
        out[f"{tag}_W"] = p.W
        out[f"{tag}_W_ai"] = p.W_ai
        out[f"{tag}_W_af"] = p.W_af
        Np = F + 2 * L
        mean = np.mean(np.abs(S))
        for ti, thr in enumerate([0.0, 0.8 * mean]):
            def fresh():
                er, ei = rl.extend(S, L, Q)
                return er, ei, np.ascontiguousarray(np.abs(er + 1j * ei))
            # batch family
            er, ei, amp = fresh()
            rl.call("LWSanyQ", er, ei, p.W, amp, F, T, L, Q, thr)
            out[f"{tag}_t{ti}_batch_any"] = er + 1j * ei
            if Q in (2, 4):
                er, ei, amp = fresh()
                rl.call(f"LWSQ{Q}", er, ei, p.W, amp, F, T, L, thr)
                out[f"{tag}_t{ti}_batch_q"] = er + 1j * ei
            # no-future family (called with W_ai like class lws does, and with W)
            for wname, Wx in (("W", p.W), ("W_ai", p.W_ai)):
                er, ei, amp = fresh()
                rl.call("NoFuture_LWSanyQ", er, ei, Wx, amp, F, T, L, Q, thr)
                out[f"{tag}_t{ti}_nofut_any_{wname}"] = er + 1j * ei
                if Q in (2, 4):
                    er, ei, amp = fresh()
                    rl.call(f"NoFuture_LWSQ{Q}", er, ei, Wx, amp, F, T, L, thr)
                    out[f"{tag}_t{ti}_nofut_q_{wname}"] = er + 1j * ei
            # asym family: (row0, M, M0) shapes used by TF_RTISI_LA plus two others
            for ai, (row0, M, M0, wname) in enumerate([(4, 1, 0, "W_ai"), (4, 1, 1, "W_af"), (1, 3, 4, "W"),
                                                       (0, T, T, "W"), (2, 5, 2, "W")]):
                Wx = {"W": p.W, "W_ai": p.W_ai, "W_af": p.W_af}[wname]
                for upd in ((2, 1) if (ti == 1 and ai in (2, 4)) else (2,)):
                    er, ei, amp = fresh()
                    rl.call("Asym_UpdatePhaseanyQ", er, ei, Wx, amp, F, M, M0, L, Q, thr, upd, row0=row0, Np=Np)
                    out[f"{tag}_t{ti}_asym{ai}_u{upd}"] = er + 1j * ei
            out[f"{tag}_asym_shapes"] = np.array([(4, 1, 0, 1), (4, 1, 1, 2), (1, 3, 4, 0), (0, T, T, 0), (2, 5, 2, 0)])
        out[f"{tag}_thr"] = np.array([0.0, 0.8 * mean])
    out["ncases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "sweeps.npz"), **out)

    # ------------------------------------------------------------------ wrapper level
    out = {}
    wcases = [(64, 16, 24, 0), (64, 32, 20, 1), (64, 8, 30, 2), (48, 16, 16, 3)]
    for fsize, fshift, T, seed in wcases:
        p = ref.lws(fsize, fshift, mode='music', batch_iterations=12, batch_alpha=3.0)
        F = fsize // 2 + 1
        tag = f"{fsize}_{fshift}"
        S = cplx(rng, (T, F))
        M = np.abs(S)
        out[f"S_{tag}"] = S
        thr = ref.get_thresholds(6, 2.0, 0.4, 1)
        out[f"thr_{tag}"] = thr
        out[f"batch_{tag}"] = ref.batch_lws(S, p.W, thr)
        out[f"batch_mag_{tag}"] = ref.batch_lws(M, p.W, thr)          # real, non-negative input
        out[f"nofuture_{tag}"] = ref.nofuture_lws(S, p.W_ai, thr[:2])
        out[f"online_{tag}"] = ref.online_lws(S, p.W, p.W_ai, p.W_af, thr[:3], 3, fshift)
        out[f"online_la0_{tag}"] = ref.online_lws(S, p.W, p.W_ai, p.W_af, thr[:3], 0, fshift)
        out[f"online_la5_{tag}"] = ref.online_lws(S, p.W, p.W_ai, p.W_af, thr[:2], 5, fshift)
        out[f"run_{tag}"] = p.run_lws(M)
        out[f"run_nofuture_{tag}"] = p.nofuture_lws(M)
        out[f"run_online_{tag}"] = p.online_lws(p.nofuture_lws(M))
        out[f"default_noop_{tag}"] = ref.lws(fsize, fshift, batch_iterations=10).run_lws(M)  # thresholds 100.. -> no-op
    np.savez_compressed(os.path.join(HERE, "wrappers.npz"), **out)

    # ------------------------------------------------------------------ general ("fractional") weights
    # The reference reads weight row N for the DC bin (lwslib.cpp:408,711,1308), one row past the end.
    # To pin defined behaviour the kernels are called directly with a tensor that has a periodic
    # extra row (row N == row 0), which is the semantics this build implements.
    out = {}
    for fsize, fshift, T in [(32, 8, 9), (32, 12, 8)]:  # integer Q=4 with general weights; fractional Q=2.67
        p = ref.lws(fsize, fshift, L=3, use_simplifications=False)
        F = fsize // 2 + 1
        Q = p.W.shape[1]
        N = fsize
        assert p.W.shape[0] == N
        tag = f"{fsize}_{fshift}"
        S = cplx(rng, (T, F))
        out[f"S_{tag}"] = S
        for wname in ("W", "W_ai", "W_af"):
            out[f"{wname}_{tag}"] = getattr(p, wname)

        def ext_w(W):
            return np.ascontiguousarray(np.concatenate([W, W[:1]], axis=0))
        thr = 0.5 * np.mean(np.abs(S))
        er, ei = rl.extend(S, 3, Q)
        amp = np.ascontiguousarray(np.abs(er + 1j * ei))
        rl.call("LWSfractionalQ", er, ei, ext_w(p.W), amp, F, T, 3, Q, thr)
        rl.call("LWSfractionalQ", er, ei, ext_w(p.W), amp, F, T, 3, Q, 0.0)
        out[f"batch_{tag}"] = er + 1j * ei
        er, ei = rl.extend(S, 3, Q)
        rl.call("NoFuture_LWSfractionalQ", er, ei, ext_w(p.W_ai), amp, F, T, 3, Q, thr)
        out[f"nofuture_{tag}"] = er + 1j * ei
        er, ei = rl.extend(S, 3, Q)
        thrs = np.ascontiguousarray(np.array([thr, 0.5 * thr]))
        w = [np.ascontiguousarray(a) for W in (p.W, p.W_ai, p.W_af) for a in (ext_w(W).real, ext_w(W).imag)]
        f = [np.ascontiguousarray(np.abs(ext_w(W)) > 1e-12, dtype=np.intc) for W in (p.W, p.W_ai, p.W_af)]
        import ctypes as C
        vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        rl.fn["TF_RTISI_LA"](vp(er), vp(ei), vp(w[0]), vp(w[1]), vp(w[2]), vp(w[3]), vp(w[4]), vp(w[5]),
                             vp(f[0]), vp(f[1]), vp(f[2]), vp(amp), 2, 2, F, T, 3, Q, float(N) / fshift, 0,
                             vp(thrs), 2)
        out[f"online_{tag}"] = er + 1j * ei
        out[f"meta_{tag}"] = np.array([fsize, fshift, T, F, Q, 3, 2])  # ..., L, LA
        out[f"thr_{tag}"] = np.array([thr])
    np.savez_compressed(os.path.join(HERE, "general_weights.npz"), **out)

    # ------------------------------------------------------------------ config-scale fingerprint (BASELINE config 2)
    # One 500 x 513 Rayleigh-magnitude spectrogram, lws(1024,256), 100 default iterations: keep the
    # input seed, summary statistics and a strided sample of the reference's output.
    p = ref.lws(1024, 256)
    g = np.random.default_rng(20260928)
    M = np.abs(g.standard_normal((500, 513)) + 1j * g.standard_normal((500, 513))).astype(np.float32).astype(np.float64)
    Y = p.run_lws(M)
    Yd = ref.batch_lws(M, p.W, np.zeros(20))
    np.savez_compressed(os.path.join(HERE, "config2_fingerprint.npz"),
                        seed=np.array(20260928), shape=np.array([500, 513]),
                        consistency_in=np.array(p.get_consistency(M.astype(complex))),
                        consistency_out=np.array(p.get_consistency(Y)),
                        norm_out=np.array(np.linalg.norm(Y)),
                        sample_idx=np.arange(0, Y.size, 97),
                        sample_out=Y.ravel()[::97],
                        consistency_dense20=np.array(p.get_consistency(Yd)),
                        sample_dense20=Yd.ravel()[::97])
    # config 1: 5 s of noise at 16 kHz, lws(512,128), literal defaults with 10 iterations = no-op
    x = np.random.default_rng(0).standard_normal(80000)
    p = ref.lws(512, 128, batch_iterations=10)
    X = p.stft(x)
    Y = p.run_lws(np.abs(X))
    p2 = ref.lws(512, 128)
    Y2 = p2.batch_lws(np.abs(X), thresholds=ref.get_thresholds(10, 1, 0.1, 1))
    np.savez_compressed(os.path.join(HERE, "config1_fingerprint.npz"),
                        shape=np.array(X.shape), noop_equal=np.array(np.array_equal(Y, np.abs(X).astype(complex))),
                        consistency_in=np.array(p.get_consistency(np.abs(X).astype(complex))),
                        consistency_out10=np.array(p2.get_consistency(Y2)),
                        sample_idx=np.arange(0, Y2.size, 53), sample_out10=Y2.ravel()[::53])
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


def extra():
    """Fingerprints of BASELINE configs 3 and 5 (added later; `python make_golden.py extra` writes only these)."""
    ref = build_reference_module()
    # config 3: run_lws of mode='music' (1 no-future sweep, 10 online iterations with look-ahead 3, 100 batch sweeps) on
    # one 500 x 513 magnitude spectrogram; the stages are kept separately
    p = ref.lws(1024, 256, mode='music')
    g = np.random.default_rng(20260928 + 3)
    M = np.abs(g.standard_normal((500, 513)) + 1j * g.standard_normal((500, 513))).astype(np.float32).astype(np.float64)
    s0 = p.nofuture_lws(M)
    s1 = p.online_lws(s0)
    s2 = p.batch_lws(s1)
    np.savez_compressed(os.path.join(HERE, "config3_fingerprint.npz"),
                        seed=np.array(20260928 + 3), shape=np.array([500, 513]),
                        consistency_nofuture=np.array(p.get_consistency(s0)),
                        consistency_online=np.array(p.get_consistency(s1)),
                        consistency_out=np.array(p.get_consistency(s2)),
                        sample_idx=np.arange(0, s0.size, 97), sample_nofuture=s0.ravel()[::97],
                        norm_online=np.array(np.linalg.norm(s1)), norm_out=np.array(np.linalg.norm(s2)))
    # config 5: 2048-point frames (1025 bins), hop 512; 150 frames, 30 sweeps of a schedule that keeps every sweep active
    p = ref.lws(2048, 512)
    g = np.random.default_rng(20260928 + 5)
    M = np.abs(g.standard_normal((150, 1025)) + 1j * g.standard_normal((150, 1025))).astype(np.float32).astype(np.float64)
    thr = ref.get_thresholds(30, 2.0, 0.1, 1)
    Y = ref.batch_lws(M, p.W, thr)
    np.savez_compressed(os.path.join(HERE, "config5_fingerprint.npz"),
                        seed=np.array(20260928 + 5), shape=np.array([150, 1025]), thr=thr,
                        consistency_in=np.array(p.get_consistency(M.astype(complex))),
                        consistency_out=np.array(p.get_consistency(Y)),
                        sample_idx=np.arange(0, Y.size, 97), sample_out=Y.ravel()[::97])
    for f in ("config3_fingerprint.npz", "config5_fingerprint.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


def fftsize():
    """stft / istft with a transform longer than the frame (lws.pyx:49-50,85,100-121) and class lws(fftsize=...), which pads its
    windows instead (lws.pyx:396-411) (added in round 5; `python make_golden.py fftsize` writes only this file)."""
    ref = build_reference_module()
    rng = np.random.default_rng(20260928 + 9)
    out = {"x": rng.standard_normal(330)}
    for fsize, nfft, hop in ((64, 128, 16), (48, 96, 16), (100, 128, 40)):
        k = f"{fsize}_{nfft}_{hop}"
        awin = np.sqrt(ref.hann(fsize, symmetric=True, use_offset=False))
        swin = ref.synthwin(awin, hop)
        out[f"awin_{k}"], out[f"swin_{k}"] = awin, swin
        for pr in (False, True):
            out[f"stft_{k}_{int(pr)}"] = ref.stft(out["x"], fsize, hop, awin, fftsize=nfft, perfectrec=pr)
        spec = rng.standard_normal((9, fsize // 2 + 1)) + 1j * rng.standard_normal((9, fsize // 2 + 1))
        try:      # the reference's istft cannot take fftsize != 2 (bins - 1): the window no longer broadcasts (lws.pyx:107-126)
            ref.istft(spec, hop, swin, fftsize=nfft)
            out[f"istft_raises_{k}"] = np.array(0)
        except ValueError:
            out[f"istft_raises_{k}"] = np.array(1)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        p = ref.lws(64, 16, fftsize=96)
    out["cls_awin_64_96_16"], out["cls_swin_64_96_16"], out["cls_W_64_96_16"] = p.awin, p.swin, p.W
    out["cls_stft_64_96_16"] = p.stft(out["x"])
    out["cls_istft_64_96_16"] = p.istft(out["cls_stft_64_96_16"])
    np.savez_compressed(os.path.join(HERE, "fftsize.npz"), **out)
    print("fftsize.npz", os.path.getsize(os.path.join(HERE, "fftsize.npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "extra":
        extra()
    elif len(sys.argv) > 1 and sys.argv[1] == "fftsize":
        fftsize()
    else:
        main()
