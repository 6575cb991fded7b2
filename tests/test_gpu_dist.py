"""GPU (-m gpu): two ranks drive the ENGINE on their shards -- the multi-GPU path of bench.py at test size.  With two or more
GPUs each rank owns one and the residual pair is reduced over RCCL (backend "nccl"); with one GPU (the test box) the two
ranks share it and the reduce goes over gloo: same sharding, same engine calls, same reduction.  Shard results must equal
the single-rank results bit for bit (spectrograms are independent; nothing is exchanged during the sweeps)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    rng = np.random.default_rng(77)
    B, T, F = 7, 90, 513
    M = np.abs(rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.float32)
    return M, B, T, F


def _worker(rank, world, port, ngpu, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import lws_amd
    from lws_amd.dist import shard_range, reduce_residual
    multi = ngpu >= world
    dev_id = rank if multi else 0
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if multi:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    M, B, T, F = _problem()
    lo, hi = shard_range(B, rank, world)
    p = lws_amd.lws(1024, 256, mode="music", batch_iterations=30, device=dev_id)
    plan = p.plan()
    t = torch.from_numpy(M[lo:hi].astype(np.complex64)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    thr = [lws_amd.get_thresholds(p.nofuture_iterations, p.nofuture_alpha, p.nofuture_beta, p.nofuture_gamma),
           lws_amd.get_thresholds(p.online_iterations, p.online_alpha, p.online_beta, p.online_gamma),
           lws_amd.get_thresholds(p.batch_iterations, p.batch_alpha, p.batch_beta, p.batch_gamma)]
    plan.run_dev(t.data_ptr(), hi - lo, T, thr[0], thr[1], p.look_ahead, 4.0, thr[2], stream=st)
    pairs = plan.residual_dev(t.data_ptr(), hi - lo, T, stream=st)
    err, pw, db = reduce_residual(pairs, device=dev if multi else None)
    q.put((rank, lo, hi, t.cpu().numpy(), pairs, err, pw, db, "nccl" if multi else "gloo"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_the_engine():
    import torch
    import torch.multiprocessing as mp
    import lws_amd
    ngpu = torch.cuda.device_count()
    assert ngpu >= 1
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ngpu, q)) for r in range(world)]
    [p.start() for p in procs]
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single rank, same engine calls on the whole batch
    M, B, T, F = _problem()
    p = lws_amd.lws(1024, 256, mode="music", batch_iterations=30)
    t = torch.from_numpy(M.astype(np.complex64)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    thr = [lws_amd.get_thresholds(p.nofuture_iterations, p.nofuture_alpha, p.nofuture_beta, p.nofuture_gamma),
           lws_amd.get_thresholds(p.online_iterations, p.online_alpha, p.online_beta, p.online_gamma),
           lws_amd.get_thresholds(p.batch_iterations, p.batch_alpha, p.batch_beta, p.batch_gamma)]
    p.plan().run_dev(t.data_ptr(), B, T, thr[0], thr[1], p.look_ahead, 4.0, thr[2], stream=st)
    pairs = p.plan().residual_dev(t.data_ptr(), B, T, stream=st)
    ref = t.cpu().numpy()
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == B
    for rank, lo, hi, out, pr, err, pw, db, backend in got:
        assert np.array_equal(out, ref[lo:hi]), "shard of rank %d differs from the single-rank result" % rank
        assert np.array_equal(pr, pairs[lo:hi])
        tot = pairs.sum(axis=0)
        assert np.isclose(err, tot[0], rtol=1e-12) and np.isclose(pw, tot[1], rtol=1e-12)
        assert backend == ("nccl" if ngpu >= world else "gloo")
    assert got[0][5:8] == got[1][5:8]          # every rank holds the same reduced pair


def test_bench_py_runs_its_multi_rank_branch():
    """bench.py's own N > 1 branch (sharding through lws_amd/dist.py, barrier + max-over-ranks timing, the residual all-reduce,
    the config-4 shard block) under the launcher the driver uses -- torch.distributed.run, 2 ranks -- on the single GPU of the
    test box: LWS_BENCH_BACKEND=gloo puts the two all-reduces on CPU tensors and lets the ranks share the GPU.  No scaling
    number is read off this (two ranks on one GPU take turns): the branch must have run, and the line must be well formed."""
    import json
    import subprocess
    env = dict(os.environ, LWS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--extra-file", os.path.join(ROOT, "gpurun_out", "bench_extra_2ranks.json")]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    assert out.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 2048      # the LAST line, compact (the driver parses it)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "shard2"
    B, T, F, it = d["config"]["batch_per_gpu"], d["config"]["frames"], d["config"]["bins"], d["config"]["iters"]
    assert np.isclose(d["value"], 2.0 * B * T * F * it / (d["ms_per_step"] * 1e-3), rtol=1e-5)   # whole job: both ranks' units
    assert d["roofline"]["frac"] > 0 and d["roofline"]["valu"]["frac_naive"] > 0
    full = json.load(open(os.path.join(ROOT, d["extra_file"])))     # everything that is not the contract's line
    ex = full["extra"]
    assert "4shard" in ex["configs"] and ex["configs"]["4shard"]["batch_per_gpu"] == 1024
    assert np.isfinite(ex["residual_db_after"]) and np.isfinite(ex["consistency_db_after"])
    assert ex["headline_checks"]["max_rel_magnitude_error"] < 1e-6


def test_bench_py_starts_its_own_ranks():
    """`python bench.py --gpus 2` exactly as the driver starts N = 1 -- no launcher, no MASTER_* / RANK / WORLD_SIZE in the
    environment: bench.py starts its two ranks itself (bench.self_launch) and the launcher's last stdout line is rank 0's JSON
    line.  One GPU here, so LWS_BENCH_BACKEND=gloo lets the ranks share it."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env["LWS_BENCH_BACKEND"] = "gloo"
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras",
           "--extra-file", os.path.join(ROOT, "gpurun_out", "bench_extra_selflaunch.json")]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    last = out.stdout.rstrip().splitlines()[-1]
    assert len([ln for ln in out.stdout.splitlines() if ln.startswith("{")]) == 1 and last.startswith("{") and len(last) < 2048
    d = json.loads(last)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["config"]["parallelism"] == "shard2"
    B, T, F, it = d["config"]["batch_per_gpu"], d["config"]["frames"], d["config"]["bins"], d["config"]["iters"]
    assert (B, T, F, it) == (256, 500, 513, 100)
    assert np.isclose(d["value"], 2.0 * B * T * F * it / (d["ms_per_step"] * 1e-3), rtol=1e-5)
    full = json.load(open(os.path.join(ROOT, d["extra_file"])))
    assert np.isfinite(full["extra"]["residual_db_after"]) and full["extra"]["headline_checks"]["max_rel_magnitude_error"] < 1e-6
    # every rank's own step time travels in the extra file (a straggler shows as min / max); the line's is the slowest rank's
    rk = full["extra"]["rank_ms_per_step"]
    assert len(rk["all"]) == 2 and rk["min"] <= rk["max"] and np.isclose(rk["max"], d["ms_per_step"], rtol=1e-6)


def test_a_rank_that_cannot_join_ends_the_job_and_is_named():
    """`python bench.py --gpus 2` with a rank that fails before the rendezvous (LWS_BENCH_FAIL_RANK, the stand-in for a GPU that does
    not open or a peer that died): the job must end with a non-zero exit code in bounded time -- not hang in init_process_group --
    and stderr must say which rank it was."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    env.update(LWS_BENCH_BACKEND="gloo", LWS_BENCH_FAIL_RANK="1", LWS_BENCH_INIT_TIMEOUT="60")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
    t0 = time.time()
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and time.time() - t0 < 120, (out.returncode, time.time() - t0)
    assert "rank 1 of 2" in out.stderr and "init_process_group" in out.stderr, out.stderr[-2000:]
    assert "rank 1 exited with code" in out.stderr                      # the launcher names it too, and stops rank 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_c_abi_residual_all_reduce_over_rccl():
    """lws_residual_allreduce_dev -- the job's residual pair for C / mex callers that run one process per GPU: this rank's sums,
    all-reduced in place on the device by RCCL (ncclAllReduce, resolved with dlopen: the library does not link against librccl).
    One GPU here, so the communicator has one rank (ncclCommInitRank through ctypes): the collective really is RCCL's, its result
    must equal the sums of lws_residual_dev; without a communicator the call is the local sum."""
    import ctypes as C
    import torch
    import lws_amd
    rng = np.random.default_rng(3)
    p = lws_amd.lws(64, 16)
    plan = p.plan()
    B, T, F = 5, 30, 33
    S = (rng.standard_normal((B, T, F)) + 1j * rng.standard_normal((B, T, F))).astype(np.complex64)
    d = torch.from_numpy(S).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    pairs = plan.residual_dev(d.data_ptr(), B, T, stream=stream)
    local = plan.residual_allreduce_dev(d.data_ptr(), B, T, comm=None, stream=stream)
    assert np.allclose(local, pairs.sum(axis=0), rtol=1e-12)
    rccl = None
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("librccl not loadable")
    class UniqueId(C.Structure):                      # ncclUniqueId: 128 bytes, passed BY VALUE to ncclCommInitRank
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        red = plan.residual_allreduce_dev(d.data_ptr(), B, T, comm=comm.value, stream=stream)
        assert np.array_equal(red, local)            # one rank: the sum over ranks is this rank's
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
