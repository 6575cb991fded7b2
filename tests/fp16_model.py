"""The fp16-complex storage mode (LWS_STORAGE_FP16) restated on the CPU: the oracle's fp64 sweeps with the state and
the target magnitudes held as IEEE half between passes over HBM, exactly where lws_systolic.hip rounds them --

  * the spectrogram is multiplied by the power of two that brings its largest fp32 magnitude to [1, 2) (store_scale; exact);
  * the input state and the target magnitudes are rounded to half once, on the way in (k_in_to_skew / k_to_skew: round to
    nearest even); the thresholds (float32(thr_i * mean|S|), those the largest magnitude exceeds) are scaled the same way and
    compared with the HALF magnitudes, and a bin is re-projected onto its HALF magnitude;
  * `nslots` consecutive effective sweeps form one pass: the state stays in fp32 (LDS rings; here fp64) inside a pass and is
    rounded to half when the last sweep of the pass writes it back;
  * on the way out only the phase comes from the half state: a bin that some sweep could have updated (half magnitude above the
    smallest effective threshold) is returned with its fp32 magnitude, every other bin as the caller gave it.

Test infrastructure (uses oracle/): tests/test_gpu_fp16.py compares the kernel with this at fp32-level bars, which a wrong
rounding mode, a wrong scale or a pass boundary in the wrong place does not meet."""
import numpy as np


def store_scale(amax):
    """2^(127 - biased exponent) of the float32 amax: what lws_systolic.hip:store_scale returns."""
    e = (np.float32(amax).view(np.uint32) >> 23) & 0xFF
    if e == 0 or e == 0xFF:
        return 1.0
    return float(np.uint32((1 if e >= 254 else 254 - int(e)) << 23).view(np.float32))


def half(x):
    return np.asarray(x, dtype=np.float64).astype(np.float16).astype(np.float64)   # round to nearest even, as v_cvt_f16_f32 / (_Float16)


def fp16_storage_batch(oracle, S, W, thresholds, nslots, round_state=True):
    """One spectrogram S (T, F) complex; thresholds relative to mean|S| as in batch_lws (lws.pyx:209-258).  Returns complex128 (T, F).
    round_state=False: the same pipeline without the per-pass rounding (must then equal the oracle's batch_lws on complex64 input)."""
    S32 = np.asarray(S).astype(np.complex64)
    W = np.ascontiguousarray(W, dtype=np.complex128)
    Qp, Q, L1 = W.shape
    L = L1 - 1
    T, F = S32.shape
    A32 = np.sqrt(S32.real.astype(np.float64) ** 2 + S32.imag.astype(np.float64) ** 2).astype(np.float32)     # mag_of
    amax = np.float32(A32.max())
    sc = store_scale(amax) if round_state else 1.0
    thr32 = (np.asarray(thresholds, dtype=np.float64) * float(A32.astype(np.float64).mean())).astype(np.float32)
    eff = [float(t) * sc for t in thr32 if amax > t]
    if not eff:
        return S32.astype(np.complex128)
    rnd = half if round_state else (lambda x: np.asarray(x, dtype=np.float64))
    er, ei = oracle.extend(S32.astype(np.complex128) * sc, L, Q)
    er, ei = np.ascontiguousarray(rnd(er)), np.ascontiguousarray(rnd(ei))
    ar, ai = oracle.extend(A32.astype(np.complex128) * sc, L, Q)             # |.| of the extended buffer = the magnitudes extended
    amp = np.ascontiguousarray(rnd(ar))
    for g0 in range(0, len(eff), nslots):
        for th in eff[g0:g0 + nslots]:
            oracle.sweep(er, ei, W, amp, F, T, L, Q, th)
        er[...], ei[...] = rnd(er), rnd(ei)                                  # the pass's write-back
    v = (er + 1j * ei)[Q - 1:Q - 1 + T, L:L + F]
    m = np.abs(v)
    touched = (m > 0) & (rnd(A32.astype(np.float64) * sc) > min(eff))
    out = S32.astype(np.complex128)
    out[touched] = v[touched] / m[touched] * A32.astype(np.float64)[touched]
    return out
