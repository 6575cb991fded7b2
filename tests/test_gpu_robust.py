"""GPU (-m gpu): behaviour at the edges of the fast path -- a failed multi-workgroup hand-over, schedules longer than one
launch's threshold table, pre-sized scratch, the device-resident pipeline entry, several shards on one node."""
import numpy as np
import pytest

import lws_amd
from lws_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fsize,fshift,storage", [(1024, 256, "fp32"), (2048, 512, "fp32"), (1024, 256, "fp16")])
def test_failed_handover_is_rerun_with_one_workgroup(fsize, fshift, storage, monkeypatch):
    """Workgroups that share a spectrogram wait for each other through HBM; if they are not resident together (shared
    or partitioned device) the wait times out.  With the poll limit forced to zero every hand-over 'fails': the call
    must still return the single-workgroup result, bit for bit, from the device-side re-run -- through the host entry
    point and through the asynchronous device one (where no host check sits between launch and result)."""
    import torch
    rng = np.random.default_rng(4)
    F = fsize // 2 + 1
    B, T = 3, 200
    S = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex64)
    thr = lws_amd.get_thresholds(30, 3.0, 0.15, 1)
    p = lws_amd.lws(fsize, fshift, storage=storage)
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "1")
    ref = p.plan().batch(S, thr)
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "3")
    assert np.array_equal(p.plan().batch(S, thr), ref)
    assert "timed out" not in p.plan().last_kernel()["name"]
    monkeypatch.setenv("LWS_SYSTOLIC_SPIN_LIMIT", "0")
    assert np.array_equal(p.plan().batch(S, thr), ref)
    assert "timed out" in p.plan().last_kernel()["name"]
    t = torch.from_numpy(S.copy()).cuda()
    p.plan().batch_dev(t.data_ptr(), B, T, thr, stream=torch.cuda.current_stream().cuda_stream)   # direct I/O path
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy().astype(np.complex128), ref)


def test_schedules_longer_than_one_launch(oracle, monkeypatch):
    """More than 440 sweeps run as several launches of the systolic kernel over the resident state (a launch boundary is
    just a longer lag between two sweeps): same result as the oracle, and as the generic engine."""
    rng = np.random.default_rng(8)
    p = lws_amd.lws(64, 16)
    S = rng.standard_normal((2, 30, 33)) + 1j * rng.standard_normal((2, 30, 33))
    thr = np.concatenate([np.full(300, 0.7), np.full(300, 0.3), np.zeros(400)])   # 1000 sweeps: 3 launches
    out = p.plan().batch(S, thr)
    info = p.plan().last_kernel()
    assert info["name"].startswith("systolic") and info["launches"] == 3, info
    pg = lws_amd.lws(64, 16, force_generic=True)
    outg = pg.plan().batch(S, thr)
    for b in range(2):
        ref = oracle.batch_lws(S[b], p.W, thr)
        assert np.linalg.norm(out[b] - ref) / np.linalg.norm(ref) < 3e-3
        assert np.linalg.norm(outg[b] - ref) / np.linalg.norm(ref) < 3e-3
    # several workgroups per spectrogram across launch boundaries: counters restart, same bits
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "1")
    ref1 = p.plan().batch(S, thr)
    monkeypatch.setenv("LWS_SYSTOLIC_NWG", "2")
    assert np.array_equal(p.plan().batch(S, thr), ref1)


def test_reserve_then_device_calls_and_run_lws_dev():
    """lws_plan_reserve pre-sizes every scratch buffer; lws_run_lws_dev == lws_run_lws on the same data."""
    import torch
    rng = np.random.default_rng(2)
    p = lws_amd.lws(1024, 256, mode="music", batch_iterations=30)
    plan = p.plan()
    B, T = 4, 90
    plan.reserve(B, T, 30)
    M = np.abs(rng.standard_normal((B, T, 513)) + 1j * rng.standard_normal((B, T, 513))).astype(np.float32)
    host = p.run_lws(M.astype(np.float64))
    t = torch.from_numpy(M.astype(np.complex64)).cuda()
    thr = [lws_amd.get_thresholds(p.nofuture_iterations, p.nofuture_alpha, p.nofuture_beta, p.nofuture_gamma),
           lws_amd.get_thresholds(p.online_iterations, p.online_alpha, p.online_beta, p.online_gamma),
           lws_amd.get_thresholds(p.batch_iterations, p.batch_alpha, p.batch_beta, p.batch_gamma)]
    plan.run_dev(t.data_ptr(), B, T, thr[0], thr[1], p.look_ahead, 4.0, thr[2], stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dev = t.cpu().numpy()
    assert np.abs(dev - host).max() < 1e-5 * M.max()
    with pytest.raises(ValueError):
        plan.reserve(0, 10, 1)


def test_multi_plan_shards_equal_one_plan():
    """lws_multi_*: contiguous blocks of the batch on one plan per device, one host thread each.  With one GPU the two
    shards share it (devices=[0, 0]) -- same code path as two GPUs; the results equal a single plan's bit for bit and the
    residual pair equals the sum of the per-spectrogram pairs."""
    rng = np.random.default_rng(6)
    p = lws_amd.lws(1024, 256, mode="music", batch_iterations=20)
    S = np.abs(rng.standard_normal((5, 60, 513)) + 1j * rng.standard_normal((5, 60, 513))).astype(np.complex128)
    thr = lws_amd.get_thresholds(20, 2.0, 0.2, 1)
    ndev = _capi.load().lws_device_count()
    devs = [0, 0] if ndev < 2 else [0, 1]
    mp = _capi.MultiPlan(513, p.W, p.W_ai, p.W_af, devices=devs)
    assert mp.shards == 2
    one = p.plan().batch(S, thr)
    assert np.array_equal(mp.batch(S, thr), one)
    t0 = lws_amd.get_thresholds(1, 1, 0.1, 1)
    t1 = lws_amd.get_thresholds(10, 1, 0.1, 1)
    assert np.array_equal(mp.run(S, t0, t1, 3, 4.0, thr), p.plan().run(S, t0, t1, 3, 4.0, thr))
    pairs = p.plan().residual(one)
    tot = mp.residual(one)
    assert np.allclose(tot, pairs.sum(axis=0), rtol=1e-12)
    mpa = _capi.MultiPlan(513, p.W)          # every visible device
    assert mpa.shards == ndev and np.array_equal(mpa.batch(S[:1], thr), one[:1])   # fewer spectrograms than shards is fine
    with pytest.raises(ValueError):
        _capi.MultiPlan(513, p.W, devices=[ndev + 3])
    mp.close(); mpa.close()


@pytest.mark.parametrize("fsize,fshift,L,T,precision", [(64, 8, 5, 40, "fp32"), (1024, 128, 5, 50, "fp32"), (48, 16, 4, 30, "fp32"),
                                                      (1040, 260, 5, 40, "fp32"), (64, 16, 7, 33, "fp64"), (1024, 128, 5, 20, "fp64"),
                                                      (1024, 256, 5, 60, "fp64"), (1024, 256, 5, 60, "fp32"), (512, 256, 5, 1, "fp64"),
                                                      (512, 256, 3, 2, "fp32"), (400, 160, 5, 45, "fp64"), (400, 160, 5, 45, "fp32"),
                                                      (2048, 512, 5, 12, "fp64"), (64, 16, 5, 200, "fp64"), (96, 32, 9, 25, "fp32"),
                                                      (4096, 1024, 5, 9, "fp32")])
def test_generic_batch_on_the_skewed_copy_is_bit_identical(fsize, fshift, L, T, precision, oracle):
    """Shapes the systolic kernels do not serve (Q = 3, L = 7, general weights, fp64 -- or here: forced) run their batch
    sweeps on the generic engine -- on a time-skewed copy of the state so that the taps of a wavefront step are coalesced.
    Same schedule and arithmetic as in the reference's layout: identical bits; and the oracle's values in fp64.  Thirteen sweeps:
    more than a group of sweeps in flight for the long frames."""
    rng = np.random.default_rng(fsize + L)
    F = fsize // 2 + 1
    p = lws_amd.lws(fsize, fshift, L=L)
    S = rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))
    thr = np.array([0.7, 0.3, 0.0, 0.0, 0.0, 0.9, 0.1, 0.0, 0.0, 0.2, 0.0, 0.0, 0.0])
    skew = _capi.Plan(F, p.W, precision=precision, force_generic=True)
    plain = _capi.Plan(F, p.W, precision=precision, force_generic=True, generic_plain_layout=True)
    a = skew.batch(S, thr)
    assert skew.last_kernel()["name"] == "generic_skew_" + precision
    b = plain.batch(S, thr)
    assert plain.last_kernel()["name"] == "generic_" + precision
    assert np.array_equal(a, b)
    if precision == "fp64":
        for i in range(2):
            assert np.abs(a[i] - oracle.batch_lws(S[i], p.W, thr)).max() < 1e-8
    skew.close(); plain.close()


@pytest.mark.parametrize("fsize,fshift,T,mode", [(1024, 256, 70, "batch"), (512, 128, 131, "batch"), (2048, 512, 40, "batch"), (1024, 128, 37, "batch"),
                                                 (1024, 512, 66, "batch"), (4096, 1024, 20, "batch"), (1000, 250, 37, "batch"), (1024, 256, 33, "music"),
                                                 (400, 160, 40, "music"), (1000, 200, 37, "batch"),
                                                 # the band engine (round 6): helper waves, a partly filled last block, Q = 16, table twiddles on long frames
                                                 (2048, 256, 70, "batch"), (1024, 64, 40, "batch"), (2000, 400, 33, "batch"), (2048, 256, 20, "music")])
def test_stale_device_memory_never_reaches_a_result(fsize, fshift, T, mode):
    """A plan's scratch comes from hipMalloc as it is; LDS is what the previous kernel left.  Whatever sits in the entries of a
    kernel's layout that no (frame, bin) owns must not reach a result, not even through a zero weight (0 x NaN): the same call on a
    fresh plan gives the same bits after the device's free memory was filled with NaN, Inf and 1e30 and given back."""
    import torch
    rng = np.random.default_rng(fsize + T)
    F = fsize // 2 + 1
    S = np.abs(rng.standard_normal((2, T, F)) + 1j * rng.standard_normal((2, T, F))).astype(np.complex128)
    S[1] *= 25.0

    def run():
        if mode == "music":
            p = lws_amd.lws(fsize, fshift, mode="music", online_iterations=3, batch_iterations=8, batch_alpha=2.0)
            return p.run_lws(S)
        p = lws_amd.lws(fsize, fshift)
        return p.plan().batch(S, lws_amd.get_thresholds(9, 2.0, 0.3, 1))
    ref = run()
    assert np.isfinite(ref).all()
    for fill in (float("nan"), float("inf"), 1e30):
        x = torch.full((1 << 28,), fill, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
        assert np.array_equal(run(), ref), fill
