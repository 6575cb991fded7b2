#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

Workload (BASELINE config 2 / SURVEY.md 8d): B=256 synthetic magnitude spectrograms per GPU, 1024-point
STFT (F=513 bins, hop 256 => Q=4, L=5), T=500 frames, 100 batch-LWS sweeps, fp32, zero initial phase.
One "step" = one complete pass of the hot path over the batch: reset the state to the magnitudes,
build the extended buffers, run the 100 in-place sweeps, extract the result -- everything resident in HBM.

The timed workload is the DENSE variant (all 100 thresholds = 0, every bin updated in every sweep) so
that no bin-iteration in the count is skipped work; the reference's default schedule 100*exp(-0.1 i)
(51.8 effective sweeps, the first ~38 are no-ops) is measured next to it and reported in "extra".

    python bench.py                       # 1 GPU, finishes in minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: spectrograms are independent, so each rank owns its own B spectrograms (weak scaling, no
data-path collective); RCCL is used only for the final consistency-residual all-reduce, outside the
timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); a copy kernel reaches 6.0-6.5 TB/s (bench extra.hbm_copy_gbs_measured)


def synth_magnitudes(B, T, F, first_seed):
    out = np.empty((B, T, F), dtype=np.float32)
    for b in range(B):
        g = np.random.default_rng(first_seed + b)
        out[b] = np.abs(g.standard_normal((T, F)) + 1j * g.standard_normal((T, F))).astype(np.float32)
    return out


def cpu_baseline(W, T, F, iters, budget_s=12.0):
    """The reference CPU path timed on this box's host cores, on a bounded sample of the same workload
    (dense thresholds).  Uses oracle/_ref (the reference's own lwslib.cpp, LWSQ4) when the prebuilt
    library travelled with the repo, else the fp64 C restatement.  Single thread = the reference's
    execution model (it never releases the GIL); an all-cores figure (one spectrogram per thread) is
    reported beside it."""
    from oracle.oracle import Oracle, RefLib, split_weights
    import ctypes as C
    orc = Oracle()
    use_ref = RefLib.available()
    rl = RefLib() if use_ref else None
    L, Q = W.shape[2] - 1, W.shape[1]
    ncores = os.cpu_count() or 1
    wr, wi, wf = split_weights(W)

    def one(seed, sweeps):
        M = synth_magnitudes(1, T, F, seed)[0].astype(np.float64)
        er, ei = orc.extend(M.astype(np.complex128), L, Q)
        amp = np.ascontiguousarray(np.abs(er + 1j * ei))
        t0 = time.perf_counter()
        for _ in range(sweeps):
            if use_ref:
                rl.fn["LWSQ4" if Q == 4 else "LWSanyQ"](
                    *( [C.c_void_p(er.ctypes.data), C.c_void_p(ei.ctypes.data), C.c_void_p(wr.ctypes.data),
                        C.c_void_p(wi.ctypes.data), C.c_void_p(wf.ctypes.data), C.c_void_p(amp.ctypes.data),
                        F, T, L] + ([] if Q == 4 else [Q]) + [0.0]))
            else:
                orc.sweep(er, ei, W, amp, F, T, L, Q, 0.0)
        return time.perf_counter() - t0

    t1 = one(1, 2)  # calibrate: two sweeps
    sweeps = int(max(4, min(iters, budget_s / 2 / (t1 / 2))))
    dt = one(2, sweeps)
    single = T * F * sweeps / dt
    # all cores: one spectrogram per thread (ctypes releases the GIL)
    times = [0.0] * ncores
    th = [threading.Thread(target=lambda i=i: times.__setitem__(i, one(10 + i, sweeps))) for i in range(ncores)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    wall = time.perf_counter() - t0
    return {"value": single, "unit": "bin*iter/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": "1 spectrogram %dx%d, %d dense sweeps, fp64, single thread" % (T, F, sweeps),
            "all_cores_value": ncores * T * F * sweeps / wall, "all_cores": ncores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="spectrograms per GPU")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-schedule", action="store_true")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--force-generic", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import lws_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    B, T, F, iters = args.batch, args.frames, 513, args.iters
    p = lws_amd.lws(1024, 256, device=local_rank, force_generic=args.force_generic)  # default sqrt-Hann pair, L=5, Q=4
    plan = p.plan()
    mags = torch.from_numpy(synth_magnitudes(B, T, F, 20260928 + rank * B)).to(dev)
    state = torch.zeros((B, T, F), dtype=torch.complex64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    thr_dense = np.zeros(iters)
    thr_default = lws_amd.get_thresholds(iters, 100, 0.1, 1)

    def step(thr):
        state.copy_(mags)  # zero phase: real, non-negative input exactly like run_lws(np.abs(X))
        plan.batch_dev(state.data_ptr(), B, T, thr, stream=stream)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def timed(thr, steps, warmup):
        for _ in range(warmup):
            step(thr)
        sync_all()
        kms, launches = 0.0, 0
        t0 = time.perf_counter()
        for _ in range(steps):
            step(thr)
            info = plan.last_kernel()  # HIP events on the launch stream; waits for this step's kernels
            kms += info["ms"]
            launches += info["launches"]
        sync_all()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, kms, launches, info["name"]

    dt, kms, launches, kname = timed(thr_dense, args.steps, args.warmup)
    ms_per_step = 1e3 * dt / args.steps
    units_per_step = float(B) * T * F * iters * world
    value = units_per_step / (dt / args.steps)

    # roofline of the dominant (update) kernel: algorithmic bytes = 20 B per active bin-iteration
    alg_bytes_per_step = 20.0 * B * T * F * iters  # per GPU
    k_ms_per_step = kms / args.steps
    achieved = alg_bytes_per_step / (k_ms_per_step * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kname,
            "kernel_ms_per_step": k_ms_per_step, "launches_per_step": launches / args.steps,
            "algorithmic_bytes_per_launch": alg_bytes_per_step / max(1.0, launches / args.steps)}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            roof["traffic"] = json.load(open(pmc)).get(kname, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass

    extra = {}
    # measured HBM copy rate of this GPU with the library's own stream-copy kernel (quote the fraction
    # against the measured copy peak as well as the spec peak); read + write bytes both counted
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        for _ in range(10):
            lws_amd._capi.check(lws_amd._capi.load().lws_stream_copy(dst.data_ptr(), src.data_ptr(), nbytes, stream or None))
        e1.record()
        torch.cuda.synchronize()
    copy_gbs = 2.0 * nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    extra["hbm_copy_gbs_measured"] = copy_gbs
    roof["frac_of_measured_copy"] = achieved / copy_gbs
    if not args.no_default_schedule:
        dt2, kms2, _, _ = timed(thr_default, max(1, args.steps), 1)
        mean = mags.mean(dim=(1, 2), keepdim=True)
        active = sum(float((mags > float(t) * mean).sum().item()) for t in thr_default)
        extra["default_schedule"] = {
            "nominal_value": units_per_step / (dt2 / max(1, args.steps)),
            "active_value": active * world / (dt2 / max(1, args.steps)),
            "effective_sweeps": active / (B * T * F),
            "ms_per_step": 1e3 * dt2 / max(1, args.steps),
            "algorithmic_GBs": (16.0 * active + 4.0 * B * T * F * iters) / (kms2 / max(1, args.steps) * 1e-3) / 1e9}

    # latency of ONE spectrogram of the same shape and schedule: its passes over HBM are pipelined over several
    # workgroups ("fewer spectrograms than CUs")
    if not args.no_config3:
        one = state[:1].clone()
        for rep in range(2):
            one.copy_(mags[:1])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            plan.batch_dev(one.data_ptr(), 1, T, thr_dense, stream=stream)
            info = plan.last_kernel()
            torch.cuda.synchronize()
            extra["single_spectrogram"] = {"wall_ms": 1e3 * (time.perf_counter() - t0), "kernel_ms": info["ms"],
                                           "kernel": info["name"]}

    # BASELINE config 3: the run_lws pipeline of lws(1024, 256, mode='music') -- 1 no-future sweep (W_ai, alpha 1),
    # 10 online iterations with look-ahead 3, 100 batch sweeps of the default schedule -- stage by stage on the device
    if not args.no_config3:
        pm = lws_amd.lws(1024, 256, mode="music", device=local_rank, force_generic=args.force_generic)
        planm = pm.plan()
        stages = [
            ("nofuture", lambda: planm.nofuture_dev(state.data_ptr(), B, T, lws_amd.get_thresholds(
                pm.nofuture_iterations, pm.nofuture_alpha, pm.nofuture_beta, pm.nofuture_gamma), wsel=1, stream=stream)),
            ("online", lambda: planm.online_dev(state.data_ptr(), B, T, lws_amd.get_thresholds(
                pm.online_iterations, pm.online_alpha, pm.online_beta, pm.online_gamma), pm.look_ahead, 4.0, stream=stream)),
            ("batch", lambda: planm.batch_dev(state.data_ptr(), B, T, lws_amd.get_thresholds(
                pm.batch_iterations, pm.batch_alpha, pm.batch_beta, pm.batch_gamma), stream=stream)),
        ]
        c3 = {}
        for rep in range(2):   # second repetition is the one reported
            state.copy_(mags)
            sync_all()
            t_all = time.perf_counter()
            for name, fn in stages:
                t0 = time.perf_counter()
                fn()
                info = planm.last_kernel()
                torch.cuda.synchronize()
                c3[name] = {"wall_ms": 1e3 * (time.perf_counter() - t0), "kernel_ms": info["ms"], "kernel": info["name"]}
            c3["total_wall_ms"] = 1e3 * (time.perf_counter() - t_all)
        c3["iterations"] = {"nofuture": pm.nofuture_iterations, "online": pm.online_iterations, "batch": pm.batch_iterations,
                            "look_ahead": pm.look_ahead}
        extra["config3_run_lws_music"] = c3

    # optional final consistency-residual reduction (the only collective): sum over all spectrograms
    step(thr_dense)
    res = plan.residual_dev(state.data_ptr(), B, T, stream=stream)
    tot = torch.tensor(res.sum(axis=0), dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    extra["residual_db_after"] = float(10 * np.log10(tot[1].item() / tot[0].item()))
    # the true consistency 20 log10(|S| / |STFT(iSTFT(S)) - S|) (lws.pyx:140-144) of the same result, on the device
    # (lws_stft.hip), summed over all spectrograms of all ranks with the same all-reduce
    t0 = time.perf_counter()
    sums = lws_amd._capi.consistency_dev(state.data_ptr(), B, T, 1024, 256, p.awin, p.swin, p.perfectrec,
                                          device=local_rank, stream=stream)
    extra["consistency_ms"] = 1e3 * (time.perf_counter() - t0)
    tot2 = torch.tensor(sums.sum(axis=0), dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot2, op=dist.ReduceOp.SUM)
    extra["consistency_db_after"] = float(10 * np.log10(tot2[0].item() / tot2[1].item()))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(p.W, T, F, iters)

    if rank == 0:
        line = {
            "metric": "complex bins*iters/sec (batch LWS, 1024-pt STFT)", "value": value, "unit": "bin*iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic Rayleigh magnitudes, numpy default_rng(20260928+b), zero phase",
            "config": {"workload": "BASELINE config 2: %d spectrograms/GPU x %d frames x %d bins, lws(1024,256) "
                                   "Q=4 L=5, %d dense batch-LWS sweeps (all thresholds 0)" % (B, T, F, iters),
                       "batch_per_gpu": B, "frames": T, "bins": F, "iters": iters, "parallelism": "shard%d" % world},
            "roofline": roof, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
