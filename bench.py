#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X, plus one roofline block per BASELINE config.

Headline (what `value` / `roofline` describe): BASELINE config 2 / SURVEY.md 8d -- B=256 synthetic magnitude
spectrograms per GPU, 1024-point STFT (F=513 bins, hop 256 => Q=4, L=5), T=500 frames, 100 batch-LWS sweeps, fp32,
zero initial phase.  One "step" = one complete pass of the hot path over the batch: reset the state to the magnitudes,
convert to the kernel's layout, run the 100 in-place sweeps, convert back -- everything resident in HBM.  The timed
workload is the DENSE variant (all thresholds 0, every bin updated in every sweep) so that no bin-iteration in the
count is skipped work; the reference's default schedule 100*exp(-0.1 i) is measured beside it (extra.default_schedule).

`extra.configs` carries one block per other BASELINE config, each with its own roofline (kernel, ms by HIP events on
the launch stream, algorithmic bytes, fraction of 8 TB/s) and the property checks of the result:
    2-T1024  the literal north-star shape 1024 x 513 (256 spectrograms of 1024 frames)
    4shard   config 4's per-GPU shard: 1024 spectrograms of 500 x 513
    3        run_lws(mode='music'): no-future -> online -> batch, stage by stage
    3-b1024  the same on 1024 spectrograms (the one-workgroup-per-spectrogram stages then run two per CU)
    3-fp64   the same pipeline on an fp64 plan (the reference's own arithmetic type), complex128 resident in HBM
    5        64 clips of 56 250 frames x 1025 bins (2048-point STFT), 200 sweeps
    5-f16    the same with fp16-complex storage (fp32 arithmetic)
    2-q2 / 2-q8   config 2's volume at hop 512 (Q = 2, the reference's LWSQ2) and hop 128 (Q = 8, LWSanyQ)
    2-f501   config 2's volume with a 1000-point window (F = 501: F-1 not a multiple of 8)
    2-f257   config 2's volume at config 1's frame size, lws(512,128): 512 spectrograms of 500 x 257
    2-q3 / 2-frac / 2-speech / 2-q5   hop = frame/3, lws(1024,384), lws(400,160), hop = frame/5: the table-twiddle builds (LWSanyQ, LWSfractionalQ)
    2-q8w / 2-q16 / 2-l8   lws(2048,256), lws(1024,64), lws(1024,256,L=8): the band engine (shapes no systolic build takes)
    host_api config 2 through the host-array entry point plan.batch(numpy complex128): what a caller of the drop-in pays
    1        BASELINE config 1: one 5 s clip (628 x 257) through lws.lws(512,128).run_lws, wall time incl. plan creation,
             beside the reference CPU path on the same clip

Every roofline block carries, beside the algorithmic-bytes HBM roofline the metric is defined on, `valu` (the arithmetic
of SURVEY 8(d) against the 157.3 TF fp32 vector peak -- what the counters say bounds these kernels) and
`hbm_measured_frac` (PMC-measured HBM bytes / kernel time / 8 TB/s).

    python bench.py                       # 1 GPU, headline + all config blocks, finishes in minutes
    python bench.py --config 5 --no-extras --steps 1      # one config as the headline (profiling passes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: spectrograms are independent, so each rank owns its own B spectrograms (weak scaling, no data-path
collective); RCCL is used only for the final consistency-residual all-reduce, outside the timed region.  Rank 0
prints ONE JSON line.
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); a copy kernel reaches 6.0-6.5 TB/s (extra.hbm_copy_gbs_measured)
VALU_PEAK_TF = 157.3    # fp32 vector peak (MI355X_MICROARCH.md)

# name -> (B per GPU, T, fsize, fshift, sweeps, storage)
BATCH_CONFIGS = {
    "2":       dict(B=256, T=500, fsize=1024, fshift=256, iters=100, storage="fp32",
                    what="BASELINE config 2: %(B)d spectrograms/GPU x %(T)d frames x %(F)d bins, lws(1024,256) Q=4 L=5"),
    "2-T1024": dict(B=256, T=1024, fsize=1024, fshift=256, iters=100, storage="fp32",
                    what="north-star shape: %(B)d spectrograms/GPU of 1024 x 513, lws(1024,256) Q=4 L=5"),
    "2-q2":    dict(B=256, T=500, fsize=1024, fshift=512, iters=100, storage="fp32",
                    what="config 2's volume at hop 512 (Q = 2, the reference's LWSQ2): %(B)d x %(T)d x %(F)d, lws(1024,512) L=5"),
    "2-q8":    dict(B=256, T=500, fsize=1024, fshift=128, iters=100, storage="fp32",
                    what="config 2's volume at hop 128 (Q = 8, the reference's LWSanyQ): %(B)d x %(T)d x %(F)d, lws(1024,128) L=5"),
    "2-f501":  dict(B=256, T=500, fsize=1000, fshift=250, iters=100, storage="fp32",
                    what="config 2's volume with a 1000-point window: %(B)d x %(T)d x %(F)d, lws(1000,250) Q=4 L=5 "
                         "(F-1 = 500 is not a multiple of 8: the frames end inside a block of the kernel's step loop)"),
    "2-f257":  dict(B=512, T=500, fsize=512, fshift=128, iters=100, storage="fp32",
                    what="config 2's volume at BASELINE config 1's frame size (16 kHz speech): %(B)d x %(T)d x %(F)d, lws(512,128) Q=4 L=5 "
                         "(short frames: the build with two sweep slots per wave)"),
    "2-q3":    dict(B=256, T=500, fsize=768, fshift=256, iters=100, storage="fp32",
                    what="hop = a third of the frame (Q = 3, the reference's LWSanyQ): %(B)d x %(T)d x %(F)d, lws(768,256) L=5 "
                         "(twiddles from a table: the build lws::tw)"),
    "2-frac":  dict(B=256, T=500, fsize=1024, fshift=384, iters=100, storage="fp32",
                    what="a hop that does not divide the frame (Qfloat = 2.67: general weights, the reference's LWSfractionalQ): "
                         "%(B)d x %(T)d x %(F)d, lws(1024,384) L=5"),
    "2-speech": dict(B=512, T=500, fsize=400, fshift=160, iters=100, storage="fp32",
                    what="25 ms frames every 10 ms at 16 kHz (Qfloat = 2.5: LWSfractionalQ): %(B)d x %(T)d x %(F)d, lws(400,160) L=5 "
                         "(table twiddles under the two-slots-per-wave build)"),
    "2-q5":    dict(B=256, T=500, fsize=1000, fshift=200, iters=40, storage="fp32",
                    what="hop = a fifth of the frame (Q = 5, LWSanyQ): %(B)d x %(T)d x %(F)d, lws(1000,200) L=5 "
                         "(the Q = 8 build's geometry with table twiddles: lws::tw_q8)"),
    # the band engine (lws_band.hip): shapes no systolic build takes
    "2-q8w":   dict(B=256, T=500, fsize=2048, fshift=256, iters=20, storage="fp32",
                    what="87.5 %% overlap at config 5's frame size (Q = 8 above 513 bins, LWSanyQ): %(B)d x %(T)d x %(F)d, lws(2048,256) L=5 "
                         "(band engine; the generic engine takes 235 ps per bin and sweep)"),
    "2-q16":   dict(B=256, T=500, fsize=1024, fshift=64, iters=10, storage="fp32",
                    what="hop = a sixteenth of the frame (Q = 16, LWSanyQ): %(B)d x %(T)d x %(F)d, lws(1024,64) L=5 (band engine)"),
    "2-l8":    dict(B=256, T=500, fsize=1024, fshift=256, iters=40, storage="fp32", L=8,
                    what="a stencil of half-width 8 (LWSQ4 with L = 8): %(B)d x %(T)d x %(F)d, lws(1024,256,L=8) (band engine)"),
    "4shard":  dict(B=1024, T=500, fsize=1024, fshift=256, iters=100, storage="fp32",
                    what="BASELINE config 4, one GPU's shard: %(B)d spectrograms x %(T)d x %(F)d, lws(1024,256)"),
    "5":       dict(B=64, T=56250, fsize=2048, fshift=512, iters=200, storage="fp32",
                    what="BASELINE config 5: %(B)d clips x %(T)d frames x %(F)d bins (10 min @ 48 kHz), lws(2048,512) Q=4 L=5"),
    "5-f16":   dict(B=64, T=56250, fsize=2048, fshift=512, iters=200, storage="fp16",
                    what="BASELINE config 5 with fp16-complex storage (fp32 arithmetic): %(B)d clips x %(T)d x %(F)d, lws(2048,512)"),
}
BYTES_ACTIVE = {"fp32": 20.0, "fp16": 10.0}    # state read + magnitude + state written, per active bin-sweep
BYTES_INACTIVE = {"fp32": 4.0, "fp16": 2.0}    # magnitude load for the threshold test


MAX_LINE = 2000   # the driver parses the LAST stdout line; round 3's 24 KB line came back as parsed = null


def _clean(x):
    """Strict JSON: NaN / Infinity become null, numpy scalars become Python numbers."""
    if isinstance(x, dict):
        return {str(k): _clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if np.isfinite(x) else None
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def _sig(x, n=6):
    return float("%.*g" % (n, x)) if isinstance(x, float) else x


def compact_roofline(roof):
    """The roofline object of the final line: what the contract names plus the two cross-checks, nothing nested deeper."""
    if not roof:
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "hbm_measured_frac",
            "frac_of_measured_copy", "kernel", "kernel_ms_per_step", "launches_per_step", "algorithmic_bytes_per_launch")
    out = {k: _sig(roof.get(k)) for k in keep if k in roof}
    # `bound` is the contract's field: the roofline achieved / peak / frac are expressed against (the metric's axis: HBM bytes,
    # repeated as `roofline_axis`); `limiter` is what the SQ counters say actually limits the kernel (vector-ALU issue or a
    # dependent chain: profiles/*pmc_sq*.json) -- the two are different questions, and for the LDS-fused kernels different answers
    out["limiter"] = out.get("bound")
    out["bound"] = "hbm"
    out["roofline_axis"] = "hbm"
    # frac is NOT a bound for the LDS-fused kernels: several sweeps run per pass over HBM, so algorithmic bytes / time can exceed
    # what HBM moves (2-q2: 1.3); the physical limit of those kernels is `limiter`, priced in `valu`
    out["frac_basis"] = "algorithmic_hbm_bytes"
    v = roof.get("valu") or {}
    out["valu"] = {"peak_tflops": v.get("peak_tflops"), "frac_naive": _sig(v.get("frac_naive"), 4), "frac_factored": _sig(v.get("frac_factored"), 4)}
    if isinstance(out.get("traffic_source"), str):
        out["traffic_source"] = out["traffic_source"][:60]
    return out


def final_line(head, world, steps, warmup, fsize, roof, cpu, extra_file=None, notes=None, also=None):
    """The ONE JSON line the driver parses: compact (< MAX_LINE bytes), strict JSON.  Everything else bench.py measures goes
    to `extra_file` and to short '# ...' lines printed before it."""
    line = {
        "metric": "complex bins*iters/sec (batch LWS, %d-pt STFT)" % fsize, "value": _sig(head["value"], 7), "unit": "bin*iter/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": _sig(head["ms_per_step"], 7),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if head["storage"] == "fp32" else "f32 math / f16 storage",
        "data": "synthetic (Rayleigh magnitudes, zero phase, random seeds 20260928+b)",
        "config": {"workload": "%d spectrograms/GPU x %d frames x %d bins, lws(%d,%d), %d %s batch-LWS sweeps, fp32, inputs resident in HBM"
                               % (head["batch_per_gpu"], head["frames"], head["bins"], fsize, head.get("fshift", 0), head["iters"], head["schedule"]),
                   "batch_per_gpu": head["batch_per_gpu"], "frames": head["frames"], "bins": head["bins"], "iters": head["iters"],
                   "schedule": head["schedule"], "parallelism": "shard%d" % world},
        "roofline": compact_roofline(roof),
        "cpu_baseline": ({k: _sig(v) for k, v in cpu.items()} if cpu else None),
    }
    if also:                                      # a few numbers of the extra blocks (each block in full: extra_file)
        line["also"] = {k: (_sig(v) if isinstance(v, float) else v) for k, v in also.items()}
    if notes:
        line["notes"] = notes
    if extra_file:
        line["extra_file"] = extra_file
    line = _clean(line)
    txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    for drop in ("notes", "also", "extra_file"):  # never let an optional field push the line over the limit
        if len(txt) >= MAX_LINE and drop in line:
            del line[drop]
            txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(txt) >= MAX_LINE:
        line["config"]["workload"] = line["config"]["workload"][:80]
        if line.get("cpu_baseline"):
            line["cpu_baseline"].pop("sample", None)
        txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(txt) < MAX_LINE, len(txt)
    return txt


def summary_lines(extra):
    """Short human-readable lines ('# name: ...', < 300 bytes each) for the blocks of extra.configs, printed before the final line."""
    out = []

    def rf(r):
        return "frac=%.3f valu=%.2f" % (r.get("frac") or 0.0, (r.get("valu") or {}).get("frac_naive") or 0.0) if r else ""
    for key in ("default_schedule",):
        b = extra.get(key)
        if b:
            out.append("# %s: %.2f ms/step, effective sweeps %.1f, %s" % (key, b["ms_per_step"], b["effective_sweeps"], rf(b.get("roofline"))))
    b = extra.get("single_spectrogram")
    if b:
        out.append("# single_spectrogram: wall %.2f ms, kernel %.2f ms (%s)" % (b["wall_ms"], b["kernel_ms"], b["kernel"]))
    for name, b in (extra.get("configs") or {}).items():
        try:
            if "error" in b or "skipped" in b:
                out.append("# %s: %s" % (name, (b.get("error") or b.get("skipped"))[:200]))
            elif name.startswith("3") and name != "3-fp64":
                out.append("# %s: total %.1f ms; " % (name, b["total_wall_ms"]) + "; ".join(
                    "%s %.2f ms (%s, %s)" % (st, b[st]["kernel_ms"], b[st]["kernel"], rf(b[st]["roofline"])) for st in ("nofuture", "online", "batch") if st in b))
            elif name == "host_api":
                out.append("# host_api: plan.batch(numpy c128 magnitudes) %.1f ms vs device-resident %.1f ms (x%.2f); complex input %s ms; run_lws_music(numpy) %.1f ms%s"
                           % (b["wall_ms"], b["device_resident_ms"], b["wall_ms"] / b["device_resident_ms"],
                              ("%.1f" % b["complex_input"]["wall_ms"]) if b.get("complex_input") else "n/a", b["run_lws_music"]["wall_ms"],
                              "".join("; %s: %.1f ms (x%.2f)" % (k, v["wall_ms"], v["ratio"]) for k, v in (b.get("other_batch_sizes") or {}).items())))
            elif name == "3-fp64":
                out.append("# 3-fp64: total %.1f ms; " % b["total_wall_ms"] + "; ".join("%s %.2f ms (%s)" % (st, b[st]["kernel_ms"], b[st]["kernel"]) for st in ("nofuture", "online", "batch") if st in b))
            elif name == "2-fp64":
                out.append("# 2-fp64: kernel %.1f ms (%s, %d launches), wall %.1f ms; generic engine %.1f ms (x%.1f)"
                           % (b["systolic"]["kernel_ms"], b["systolic"]["kernel"], b["systolic"]["launches"] or 0, b["systolic"]["wall_ms"],
                              b["generic"]["kernel_ms"], b["speedup_over_generic"]))
            elif name == "1":
                out.append("# 1: " + "; ".join("%s %.2f ms (cpu %.1f ms, rel-L2 %.1e)" % (k, v["wall_ms"], v["cpu_reference_ms"], v["checks"]["rel_l2_vs_cpu"])
                                                for k, v in b.items() if isinstance(v, dict) and "wall_ms" in v))
            elif "roofline" in b:
                r = b["roofline"]
                out.append("# %s: %.2f ms/step, kernel %.2f ms (%s), %s%s%s" % (name, b["ms_per_step"], r["kernel_ms_per_step"], r["kernel"], rf(r),
                                                                              (", %.1f ps per bin-sweep" % b["kernel_ps_per_bin_sweep"]) if "kernel_ps_per_bin_sweep" in b else "",
                                                                              (", default schedule %.1f ms" % b["default_schedule"]["ms_per_step"]) if "default_schedule" in b else ""))
        except Exception as e:       # a summary must never cost the final line
            out.append("# %s: (summary failed: %s)" % (name, e))
    return [l[:300] for l in out]


def synth_magnitudes(B, T, F, first_seed):
    out = np.empty((B, T, F), dtype=np.float32)
    for b in range(B):
        g = np.random.default_rng(first_seed + b)
        out[b] = np.abs(g.standard_normal((T, F)) + 1j * g.standard_normal((T, F))).astype(np.float32)
    return out


def device_magnitudes(torch, dev, B, T, F, first_seed):
    """Rayleigh magnitudes of the SURVEY 8(d) generator for shapes numpy handles in seconds; generated on the
    device (torch Philox, same distribution) for config 5, whose 3.7e9 values would take numpy minutes."""
    if B * T * F <= (1 << 29):
        return torch.from_numpy(synth_magnitudes(B, T, F, first_seed)).to(dev), "numpy default_rng(20260928+b)"
    g = torch.Generator(device=dev)
    g.manual_seed(first_seed)
    out = torch.empty((B, T, F), dtype=torch.float32, device=dev)
    for b in range(B):   # one clip at a time: no 2x temporary of the whole batch
        re = torch.randn((T, F), generator=g, device=dev)
        im = torch.randn((T, F), generator=g, device=dev)
        out[b] = torch.sqrt(re * re + im * im)
    return out, "torch.randn on the device (Philox, seed 20260928)"


def physical_cores():
    """(hardware threads this process may run on, physical cores among them, CPUs the container's quota allows).
    Hyper-threads share a core's FPU: one spectrogram per core; a cgroup CPU quota (cpu.max: the GPU boxes of this pool give a
    container 16 CPUs of their 256 hardware threads) caps what any number of threads can use."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen = set()
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                seen.add(f.read().strip())
        except OSError:
            seen.add(str(c))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                    # cgroup v2: "<quota us> <period us>" or "max <period>"
            q, per = f.read().split()[:2]
            quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # v1
                q, per = float(f.read()), float(g.read())
                quota = q / per if q > 0 else None
        except (OSError, ValueError):
            quota = None
    return len(allowed), max(1, len(seen)), quota


def cpu_baseline(W, T, F, iters, budget_s=12.0, plan=None):
    """The reference CPU path timed on this box's host cores, on a bounded sample of the same workload
    (dense thresholds).  Uses oracle/_ref (the reference's own lwslib.cpp, LWSQ4) when the prebuilt
    library travelled with the repo, else the fp64 C restatement.  Single thread = the reference's
    execution model (it never releases the GIL); an all-cores figure (one spectrogram per thread) is
    reported beside it: every thread's inputs are built before the clock starts, the threads leave a barrier
    together and only their sweeps (ctypes calls, GIL released) are timed.
    With `plan` (the plan that was just timed): the parity of the timed arithmetic path, measured now -- one T x F
    spectrogram from a random-phase start, `iters` dense sweeps on the GPU against the same CPU sweeps."""
    from oracle.oracle import Oracle, RefLib, split_weights
    import ctypes as C
    orc = Oracle()
    use_ref = RefLib.available()
    rl = RefLib() if use_ref else None
    L, Q = W.shape[2] - 1, W.shape[1]
    nthreads, pcores, quota = physical_cores()
    ncores = max(1, min(pcores, int(quota) if quota else pcores))     # threads of the all-cores figure: what can actually run at once
    wr, wi, wf = split_weights(W)

    def prepare(S):
        er, ei = orc.extend(np.asarray(S, dtype=np.complex128), L, Q)
        return er, ei, np.ascontiguousarray(np.abs(er + 1j * ei))

    wfi = np.ascontiguousarray(wf, dtype=np.intc)
    ref_fn = C.cast(rl.fn["LWSQ4" if Q == 4 else "LWSanyQ"], C.c_void_p) if use_ref else None

    def sweep(er, ei, amp, sweeps):
        """`sweeps` dense sweeps in ONE C call (oracle/lws_oracle.c: lwso_repeat_*): the timed region never enters the interpreter"""
        if use_ref:   # the reference's own LWSQ4 / LWSanyQ (lwslib.cpp:153-373)
            orc.lib.lwso_repeat_kernel(ref_fn, 0 if Q == 4 else 1, er.ctypes.data, ei.ctypes.data, wr.ctypes.data, wi.ctypes.data, wfi.ctypes.data,
                                       amp.ctypes.data, F, T, L, Q, 0.0, sweeps)
        else:
            orc.lib.lwso_repeat_sweep(er.ctypes.data, ei.ctypes.data, wr.ctypes.data, wi.ctypes.data, wfi.ctypes.data, amp.ctypes.data,
                                      F, T, L, Q, W.shape[0], 0.0, sweeps)

    def one(seed, sweeps):
        bufs = prepare(synth_magnitudes(1, T, F, seed)[0].astype(np.float64))
        t0 = time.perf_counter()
        sweep(*bufs, sweeps)
        return time.perf_counter() - t0

    t1 = one(1, 2)  # calibrate: two sweeps
    sweeps = int(max(4, min(iters, budget_s / 2 / (t1 / 2))))
    dt = one(2, sweeps)
    single = T * F * sweeps / dt
    # all cores: one spectrogram per physical core; inputs first, then everybody leaves the barrier and only the sweeps are timed
    bufs = [prepare(synth_magnitudes(1, T, F, 10 + i)[0].astype(np.float64)) for i in range(ncores)]
    wall = None
    for _attempt in range(2):   # the better of two rounds: the first one also pays for waking the cores and starting the threads
        gate = threading.Barrier(ncores + 1)
        ends = [0.0] * ncores

        def work(i):
            sweep(*bufs[i], 1)      # (untimed: the thread's core awake, its buffers in its caches)
            gate.wait()
            sweep(*bufs[i], sweeps)
            ends[i] = time.perf_counter()

        th = [threading.Thread(target=work, args=(i,)) for i in range(ncores)]
        [t.start() for t in th]
        gate.wait()
        t0 = time.perf_counter()
        [t.join() for t in th]
        wall = max(ends) - t0 if wall is None else min(wall, max(ends) - t0)
    out = {"value": single, "unit": "bin*iter/s", "cores": 1, "kind": "reference" if use_ref else "port",
           "sample": "1 spectrogram %dx%d, %d dense sweeps, fp64, single thread" % (T, F, sweeps),
           "all_cores_value": ncores * T * F * sweeps / wall, "all_cores": ncores, "hw_threads": nthreads, "physical_cores": pcores,
           "cpu_quota": quota}
    if plan is not None:
        # three spectrograms (the tail of such a comparison -- bins whose weighted sum nearly cancels, where fp32 rounding decides the
        # phase -- differs from input to input: 0 to ~60 bins of 256 500), aggregated
        NP = 3
        rng = np.random.default_rng(20260929)
        S = np.stack([(synth_magnitudes(1, T, F, 3 + i)[0] * np.exp(2j * np.pi * rng.random((T, F)))).astype(np.complex64).astype(np.complex128) for i in range(NP)])
        got = plan.batch(S, np.zeros(iters))
        kname = plan.last_kernel()["name"]
        ref = np.empty_like(S)
        for i in range(NP):
            er, ei, amp = prepare(S[i])
            sweep(er, ei, amp, iters)
            ref[i] = (er + 1j * ei)[Q - 1:Q - 1 + T, L:L + F]
        mean = float(np.mean(np.abs(S)))

        def figures(x):
            d = np.abs(x - ref)
            return {"rel_l2": float(np.linalg.norm(x - ref) / np.linalg.norm(ref)), "median_over_mean": float(np.median(d) / mean),
                    "p999_over_mean": float(np.quantile(d, 0.999) / mean), "max_over_mean": float(d.max() / mean),
                    "bins_off_by_1e-2_mean": int((d > 1e-2 * mean).sum()), "bins": int(d.size)}
        out["parity"] = figures(got)
        out["parity"]["what"] = ("%d x %dx%d, %d dense sweeps from random phases: timed plan (%s) vs %s, measured in this run"
                                 % (NP, T, F, iters, kname, "oracle/_ref LWSQ4" if use_ref and Q == 4 else ("oracle/_ref" if use_ref else "oracle")))
        try:
            # the same input through the order-exact fp32 engine (the reference's own order of operations in fp32): what fp32 state costs
            # whatever the kernel -- the figure the timed kernel's is to be read against
            from lws_amd import _capi
            gen = _capi.Plan(F, W, force_generic=True)
            out["parity"]["order_exact_fp32"] = figures(gen.batch(S, np.zeros(iters)))
            gen.close()
        except Exception as e:   # (the comparison is context, not the measurement)
            out["parity"]["order_exact_fp32"] = {"error": str(e)[:100]}
        if use_ref:
            # ... and how far the fp64 reference is from ITSELF on this input: the same sweeps in the oracle's canonical form (LWSanyQ's
            # grouping instead of LWSQ4's, lwslib.cpp:153-373), a re-association of fp64 sums
            alt = np.empty_like(S)
            for i in range(NP):
                er2, ei2, amp2 = prepare(S[i])
                orc.lib.lwso_repeat_sweep(er2.ctypes.data, ei2.ctypes.data, wr.ctypes.data, wi.ctypes.data, wfi.ctypes.data, amp2.ctypes.data,
                                          F, T, L, Q, W.shape[0], 0.0, iters)
                alt[i] = (er2 + 1j * ei2)[Q - 1:Q - 1 + T, L:L + F]
            out["parity"]["fp64_reassociation"] = figures(alt)
    return out


def flops_per_active_bin(W):
    """Arithmetic of one bin update: one complex multiply-add (8 flop) per tap whose weight is non-zero --
    a weight W[r][k] serves the taps (m-+r, c-+k): 4 of them, 2 if r or k is 0 -- plus the re-projection (|.|^2, rsqrt,
    scale: ~10).  `naive`: every tap its own multiply-add, as LWSanyQ does (lwslib.cpp:283-373); `factored`: the grouped form
    of LWSQ2 / LWSQ4 (lwslib.cpp:123-128), 3/4 of it.  lws(1024,256): 60 taps -> 490 / 370 flop."""
    nz = np.abs(np.asarray(W)[0]) > 1e-12          # [r][k], the same for every row of the weights
    taps = 0
    for r in range(nz.shape[0]):
        for k in range(nz.shape[1]):
            if nz[r, k] and (r or k):
                taps += 4 if (r and k) else 2
    return {"taps": int(taps), "naive": 8.0 * taps + 10.0, "factored": 6.0 * taps + 10.0}


def roofline_block(alg_bytes, k_ms, active_bins, W, traffic, tsrc, kernel, launches, bound):
    """The roofline object of one kernel: `achieved/peak/frac` are the algorithmic-bytes HBM roofline BASELINE.json's metric
    is defined on; `valu` prices the same launch's arithmetic against the fp32 vector peak; `hbm_measured_frac` is what the
    PMC passes saw go through HBM.  `bound` names the limit the counters point at (profiles/r0*_pmc_sq*.json)."""
    sec = k_ms * 1e-3
    ach = alg_bytes / sec / 1e9
    fl = flops_per_active_bin(W)
    blk = {"bound": bound, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": traffic, "traffic_source": tsrc,
           "hbm_measured_frac": (traffic / sec / 1e9 / HBM_PEAK_GBS) if traffic else None,
           "valu": {"peak_tflops": VALU_PEAK_TF, "flop_per_active_bin_naive": fl["naive"], "flop_per_active_bin_factored": fl["factored"],
                    "achieved_tflops_naive": fl["naive"] * active_bins / sec / 1e12, "achieved_tflops_factored": fl["factored"] * active_bins / sec / 1e12,
                    "frac_naive": fl["naive"] * active_bins / sec / 1e12 / VALU_PEAK_TF, "frac_factored": fl["factored"] * active_bins / sec / 1e12 / VALU_PEAK_TF},
           "kernel": kernel, "kernel_ms_per_step": k_ms, "launches_per_step": launches,
           "algorithmic_bytes_per_launch": alg_bytes / max(1.0, launches)}
    return blk


def _source_hashes():
    """sha1 of the kernel sources next to the running library (they travel with it): a committed PMC profile names the ones it was
    taken on (tools/collect_profiles.py), so that a kernel change without tools/profile.sh does not report stale bytes silently."""
    import hashlib
    out = {}
    d = os.path.join(ROOT, "lws_amd", "csrc")
    for fn in ("lws_systolic.hip", "lws_online.hip", "lws_nofuture.hip", "lws_common.h", "lws_band.hip", "lws_band_core.h"):
        try:
            out[fn] = hashlib.sha1(open(os.path.join(d, fn), "rb").read()).hexdigest()
        except OSError:
            pass
    return out


_PROFILE_KERNEL_OF = {"systolic": "k_systolic", "online": "k_online", "nofuture": "k_nofuture", "band": "k_band<"}


def traffic_entry_matches(ent, kname):
    """Is the profile entry `ent` a measurement of the kernel the plan just ran (`kname` = lws_last_kernel_name)?  Entries written
    since round 5 carry that name (engine_kernel); older ones only the profiler's C++ name, checked by kernel family."""
    if not ent or not kname:
        return False
    if ent.get("engine_kernel"):
        return ent["engine_kernel"] == kname
    fam = next((v for k, v in _PROFILE_KERNEL_OF.items() if str(kname).startswith(k)), None)
    return bool(fam) and fam in str(ent.get("kernel", ""))


def load_traffic(kname, config, stage=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*pmc_traffic*.json): NOT measured in
    this run -- PMC collection needs the profiler around the process (tools/profile.sh regenerates the files).  An entry
    recorded for another kernel than the one that just ran is refused (traffic = null, the reason in traffic_source); one whose
    kernel source has changed since is returned marked STALE."""
    refused = None
    for fn in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
        except Exception:
            continue
        ent = d.get("configs", {}).get(config) or (d.get(kname) if config == "2" else None)
        if ent and stage:
            ent = ent.get(stage)
        if ent and ent.get("hbm_bytes_per_launch"):
            if "kernel" in ent and not traffic_entry_matches(ent, kname):
                refused = refused or "refused profiles/%s: taken on %s, this run's kernel is %s" % (fn, ent.get("engine_kernel") or str(ent.get("kernel"))[:60], kname)
                continue
            src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed; not this run)" % fn
            rec = d.get("_sources") or {}
            now = _source_hashes()
            # (only the sources of the kernel that ran: a change of the band engine does not age the systolic kernel's figure)
            mine = {"systolic": ("lws_systolic.hip", "lws_common.h"), "online": ("lws_online.hip", "lws_common.h"), "nofuture": ("lws_nofuture.hip", "lws_common.h"),
                    "band": ("lws_band.hip", "lws_band_core.h", "lws_common.h")}
            keep = next((v for k, v in mine.items() if str(kname).startswith(k)), None)
            changed = sorted(k for k in rec if k in now and rec[k] != now[k] and (keep is None or k in keep))
            if changed:
                src = "STALE (%s changed since) " % ",".join(changed) + src
            return ent["hbm_bytes_per_launch"], src
    return None, refused


def measure_traffic(config, kname, timeout_s=300):
    """HBM bytes per step of the update kernel, MEASURED NOW: two rocprofv3 passes (--pmc FETCH_SIZE, then --pmc WRITE_SIZE: separate
    passes, with --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) around a child `bench.py --config <config>
    --no-extras --steps 1 --warmup 1` of this very workload, read back from the rocpd database: (2 FETCH_SIZE + WRITE_SIZE) KiB of the
    timed step's launches of the kernel the plan ran (gfx950 tallies 128-B fetches as 64 B: doubled; WRITE_SIZE is exact).  Runs after
    the timed region, outside it.  Returns (bytes, source) or (None, why not) -- the caller then falls back to the committed profile."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    fam = next((v for k, v in _PROFILE_KERNEL_OF.items() if str(kname).startswith(k)), None)
    if not fam:
        return None, "no profiler kernel name known for %s" % kname
    work = tempfile.mkdtemp(prefix="lws_traffic_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, ctr)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "-d", out, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--config", config,
                   "--no-extras", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-traffic-pass", "--extra-file", os.path.join(work, "extra.json")]
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
                env.pop(k, None)
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (ctr, r.returncode)
            con = sqlite3.connect(dbs[0])
            tabs = [t[0] for t in con.execute("select name from sqlite_master where type in ('table','view')")]
            ct = next((t for t in tabs if t.startswith("counters_collection")), None)
            if not ct:
                return None, "no counters in the rocpd database"
            cols = [c[1] for c in con.execute("pragma table_info(%s)" % ct)]
            order = "dispatch_id" if "dispatch_id" in cols else "rowid"
            v = [x[0] for x in con.execute("select value from %s where counter_name=? and kernel_name like ? order by %s" % (ct, order), (ctr, "%" + fam + "%"))]
            con.close()
            if not v:
                return None, "no dispatch of %s in the profile" % fam
            # the child ran a warm-up step and the timed step with the same number of launches: the step is the second half
            per_step = max(1, len(v) // 2)
            vals[ctr] = float(sum(v[-per_step:]))
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes around a child run of "
                                                                            "this workload (2 x FETCH + WRITE, KiB)")
    except Exception as e:       # (a profiler problem must never cost the line)
        return None, "traffic pass failed: %s: %s" % (type(e).__name__, str(e)[:120])
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _device_count():
    """GPUs this process could open, without initialising a HIP context in the launcher (the ranks are fresh processes)."""
    try:
        import lws_amd
        return int(lws_amd._capi.load().lws_device_count())
    except Exception:
        import torch
        return int(torch.cuda.device_count())


def self_launch(n, argv, script=None, have=None):
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this file (one process per GPU) with the
    environment torch.distributed.run would give them -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free
    MASTER_PORT -- and wait for them.  Rank 0 inherits stdout, so its ONE JSON line is this process's last line.  Under RCCL a
    rank needs its own GPU: with fewer GPUs than N the job runs on the GPUs there are (said on stderr; the line's n_gpus is the
    number of ranks that ran).  LWS_BENCH_BACKEND=gloo lets ranks share a GPU (the tests' two ranks on a one-GPU box).
    Returns the exit code."""
    import subprocess
    backend = os.environ.get("LWS_BENCH_BACKEND", "nccl")
    script = script or os.path.abspath(__file__)
    have = _device_count() if have is None else have
    if have < 1:
        print("bench.py: no GPU visible to liblws_hip (hipGetDeviceCount = %d): nothing to measure" % have, file=sys.stderr)
        return 2
    world = n
    if backend == "nccl" and have < n:
        print("# bench.py: --gpus %d asked, %d GPU(s) visible: running %d rank(s) (RCCL needs one GPU per rank)" % (n, have, have), file=sys.stderr)
        world = have
    # the children's --gpus is the world that actually runs
    child_argv, skip = [], False
    for i, a in enumerate(argv):
        if skip:
            skip = False
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus="):
            continue
        child_argv.append(a)
    child_argv = ["--gpus", str(world)] + child_argv
    env = dict(os.environ, WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs between the processes of this host
    if world == 1:
        for k in ("WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
            env.pop(k)
        return subprocess.call([sys.executable, script] + child_argv, env=env)
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, script] + child_argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))   # same process group as this launcher
    rc = 0
    try:
        live = set(range(world))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is None:
                    continue
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, c), file=sys.stderr)
                    for q in sorted(live):        # our own children, by PID
                        procs[q].terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for pr in procs:
            if pr.poll() is None:
                pr.terminate()
        rc = 130
    return rc


def pin_rank_to_numa(torch, local_rank, local_world):
    """Best effort: bind this rank's host threads (the host-array path runs up to 32 workers per plan) to the cores of its GPU's
    NUMA node -- /sys/bus/pci/devices/<bus id>/numa_node -- or, when the node is unknown (-1), to an even 1/local_world slice of
    the cores this process may use.  Only for N > 1; LWS_BENCH_NO_PIN=1 switches it off.  Returns a description or None."""
    if local_world <= 1 or os.environ.get("LWS_BENCH_NO_PIN") or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cpus, how = None, None
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id"), getattr(pr, "pci_bus_id"), getattr(pr, "pci_device_id"))
            node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
            if node >= 0:
                txt = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
                got = set()
                for part in txt.split(","):
                    a, _, b = part.partition("-")
                    got.update(range(int(a), int(b or a) + 1))
                got &= set(allowed)
                # ranks whose GPUs share the node split its cores between them
                same = [i for i in range(local_world) if _numa_of(torch, i) == node]
                if got and local_rank in same:
                    got = sorted(got)
                    k, j = len(same), same.index(local_rank)
                    per = max(1, len(got) // k)
                    cpus, how = got[j * per:(j + 1) * per] or got, "numa node %d (%s)" % (node, bus)
        except Exception:
            cpus = None
        if not cpus:
            per = max(1, len(allowed) // local_world)
            cpus, how = allowed[local_rank * per:(local_rank + 1) * per] or allowed, "even slice"
        os.sched_setaffinity(0, cpus)
        return "%d cores, %s" % (len(cpus), how)
    except Exception:
        return None


def _numa_of(torch, i):
    try:
        pr = torch.cuda.get_device_properties(i)
        bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        return int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
    except Exception:
        return -1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="2", choices=sorted(BATCH_CONFIGS), help="headline workload (default: BASELINE config 2)")
    ap.add_argument("--batch", type=int, default=None, help="spectrograms per GPU (default: the config's)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--schedule", default="dense", choices=["dense", "default"], help="thresholds of the headline workload")
    ap.add_argument("--extras", default=None, help="comma list of config blocks for extra.configs (default: all at N=1, 4shard at N>1)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-default-schedule", action="store_true")
    ap.add_argument("--no-traffic-pass", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child runs (the committed profile is quoted instead)")
    ap.add_argument("--force-generic", action="store_true")
    ap.add_argument("--extra-file", default=None, help="where the full (non-contract) results go (default: gpurun_out/bench_extra.json if "
                                                       "that directory exists, else ./bench_extra.json)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N` (how the driver starts N = 1): start the N ranks here, one process per GPU
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    import lws_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        print("# bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs (n_gpus = %d)" % (args.gpus, world, world),
              file=sys.stderr)
    # RCCL ("nccl") between the GPUs of a node.  LWS_BENCH_BACKEND=gloo runs the same N > 1 code path with the (two)
    # all-reduces on CPU tensors, and lets ranks share a GPU when there are fewer GPUs than ranks: tests/test_gpu_dist.py
    # drives this file with 2 ranks on the single GPU of the test box.
    backend = os.environ.get("LWS_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pinned = pin_rank_to_numa(torch, local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that cannot join (a GPU that does not open, a peer that died before the rendezvous) must end the job with a
        # non-zero exit code in bounded time and say who it is: the rendezvous has a deadline (LWS_BENCH_INIT_TIMEOUT seconds,
        # default 180), a failure is reported with the rank, and the launcher -- self_launch below, or torch.distributed.run --
        # stops the other ranks when one exits non-zero.  (LWS_BENCH_FAIL_RANK=<r>: rank r fails here on purpose -- tests.)
        import datetime
        try:
            if os.environ.get("LWS_BENCH_FAIL_RANK") == str(rank):
                raise RuntimeError("LWS_BENCH_FAIL_RANK=%d: failing before the rendezvous (test hook)" % rank)
            deadline = datetime.timedelta(seconds=int(os.environ.get("LWS_BENCH_INIT_TIMEOUT", "180")))
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, timeout=deadline)
            else:
                dist.init_process_group(backend, timeout=deadline)
        except Exception as e:
            print("bench.py: rank %d of %d (GPU %d): init_process_group(%s) failed: %s: %s" % (rank, world, local_rank, backend, type(e).__name__, str(e)[:300]),
                  file=sys.stderr)
            sys.stderr.flush()
            os._exit(3)       # (not sys.exit: a half-initialised process group must not get to run its destructors against dead peers)
    from lws_amd.dist import reduce_residual, shard_range
    stream = torch.cuda.current_stream().cuda_stream
    have_f16 = hasattr(lws_amd._capi, "LWS_STORAGE_FP16")

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    plans = {}

    def engine(fsize, fshift, storage="fp32", mode=None, L=5):
        key = (fsize, fshift, storage, mode, L)
        if key not in plans:
            kw = dict(device=local_rank, force_generic=args.force_generic, L=L)
            if storage == "fp16":
                kw["storage"] = "fp16"
            if mode:
                kw["mode"] = mode
            plans[key] = lws_amd.lws(fsize, fshift, **kw)
        return plans[key]

    def run_batch_config(name, steps, warmup, schedule="dense", B=None, T=None, iters=None, checks=True):
        """Time `steps` passes of one batch-LWS configuration; returns the block with its roofline."""
        cfg = dict(BATCH_CONFIGS[name])
        B = B or cfg["B"]; T = T or cfg["T"]; iters = iters or cfg["iters"]
        F = cfg["fsize"] // 2 + 1
        storage = cfg["storage"]
        p = engine(cfg["fsize"], cfg["fshift"], storage, L=cfg.get("L", 5))
        plan = p.plan()
        lo, _hi = shard_range(B * world, rank, world)      # this rank's contiguous block of the job's B * world spectrograms
        mags, gen = device_magnitudes(torch, dev, B, T, F, 20260928 + lo)
        state = torch.empty((B, T, F), dtype=torch.complex64, device=dev)
        thr = np.zeros(iters) if schedule == "dense" else lws_amd.get_thresholds(iters, 100, 0.1, 1)

        def step(th):
            state.copy_(mags)  # zero phase: real, non-negative input exactly like run_lws(np.abs(X))
            plan.batch_dev(state.data_ptr(), B, T, th, stream=stream)

        for i in range(warmup):
            step(thr if B * T * F <= (1 << 29) else thr[:min(3, iters)])   # big shapes: a short call allocates the same scratch
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
        rank_ms = None
        if world > 1:
            # the step time of the job is the slowest rank's; every rank's own is kept too (extra: a straggler shows in min / max)
            mine = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rank_ms = [1e3 * float(x.item()) / steps for x in every]
            dt = max(float(x.item()) for x in every)
        units = float(B) * T * F * iters
        if schedule == "dense":
            active = units
        else:
            mean = mags.mean(dim=(1, 2), keepdim=True)
            active = sum(float((mags > float(t) * mean).sum().item()) for t in thr)
        alg = (BYTES_ACTIVE[storage] - BYTES_INACTIVE[storage]) * active + BYTES_INACTIVE[storage] * units   # per GPU
        k_ms = kms / steps
        traffic, tsrc = load_traffic(info["name"], name)
        blk = {
            "workload": (cfg["what"] % dict(B=B, T=T, F=F)) + ", %d %s batch-LWS sweeps" % (iters, "dense (all thresholds 0)" if schedule == "dense" else "default-schedule (100 exp(-0.1 i))"),
            "batch_per_gpu": B, "frames": T, "bins": F, "iters": iters, "schedule": schedule, "storage": storage,
            "data": "synthetic Rayleigh magnitudes, %s, zero phase" % gen,
            "steps": steps, "ms_per_step": 1e3 * dt / steps, "value": units * world / (dt / steps), "kernel_ps_per_bin_sweep": 1e9 * k_ms / units,
            "rank_ms_per_step": ({"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms} if rank_ms else None),
            "active_value": active * world / (dt / steps), "effective_sweeps": active / (float(B) * T * F),
            # bound: the vector ALU (66 % busy, HBM at 0.16x the algorithmic bytes: profiles/r02_pmc_sq_counters.json)
            "roofline": roofline_block(alg, k_ms, active, p.W, traffic, tsrc, info["name"], launches / steps,
                                       "valu" if str(info["name"]).startswith(("systolic", "band")) else "latency"),
        }
        if checks:
            # size-independent properties of the result (the -m gpu tests assert the same ones at these sizes)
            n_chk = min(B, 4)
            out = state[:n_chk]
            mg = mags[:n_chk]
            tol = 1e-6   # (fp16 storage too: the output takes its magnitudes from the fp32 targets, only the phase from the fp16 state)
            blk["checks"] = {"max_rel_magnitude_error": float(((out.abs() - mg).abs().max() / mg.max()).item()), "magnitude_tolerance": tol,
                             "finite": bool(torch.isfinite(torch.view_as_real(state)).all().item())}
            if cfg["fsize"] <= 2048:
                c0 = lws_amd._capi.consistency_dev(mags[:1].to(torch.complex64).data_ptr(), 1, T, cfg["fsize"], cfg["fshift"], p.awin, p.swin,
                                                   p.perfectrec, device=local_rank, stream=stream)
                c1 = lws_amd._capi.consistency_dev(out.data_ptr(), 1, T, cfg["fsize"], cfg["fshift"], p.awin, p.swin, p.perfectrec,
                                                   device=local_rank, stream=stream)
                blk["checks"]["consistency_db_before"] = float(10 * np.log10(c0[0, 0] / c0[0, 1]))
                blk["checks"]["consistency_db_after"] = float(10 * np.log10(c1[0, 0] / c1[0, 1]))
        return blk, (p, plan, mags, state)

    # ---- headline ---------------------------------------------------------------------------------------------------
    head, (p, plan, mags, state) = run_batch_config(args.config, args.steps, args.warmup, schedule=args.schedule,
                                                    B=args.batch, T=args.frames, iters=args.iters)
    B, T, F, iters = head["batch_per_gpu"], head["frames"], head["bins"], head["iters"]
    roof = dict(head["roofline"])
    extra = {"headline_checks": head.get("checks")}
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_", "ROCTX")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")
    if world == 1 and rank == 0 and not args.no_traffic_pass and not under_profiler and not args.force_generic and args.batch is None and args.frames is None and args.iters is None \
            and args.schedule == "dense":
        # roofline.traffic measured in this run (two rocprofv3 --pmc passes around a child run of the headline workload: ~1 min);
        # the committed profile's figure stays in the extra file beside it
        got, how = measure_traffic(args.config, roof["kernel"])
        extra["traffic_pass"] = {"bytes": got, "how": how, "committed_profile": {"bytes": roof.get("traffic"), "source": roof.get("traffic_source")}}
        if got:
            sec = roof["kernel_ms_per_step"] * 1e-3
            roof["traffic"], roof["traffic_source"] = got, how
            roof["hbm_measured_frac"] = got / sec / 1e9 / HBM_PEAK_GBS
    if head.get("rank_ms_per_step"):
        extra["rank_ms_per_step"] = head["rank_ms_per_step"]     # every rank's own step time: a straggler shows as min / max
    if pinned:
        extra["host_affinity_rank0"] = pinned

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
    roof["frac_of_measured_copy"] = roof["achieved"] / copy_gbs

    # optional final consistency-residual reduction (the only collective): sum over all spectrograms of all ranks
    if B * T * F <= (1 << 29):
        res = plan.residual_dev(state.data_ptr(), B, T, stream=stream)
        _err, _pw, extra["residual_db_after"] = reduce_residual(res, device=coll_dev)   # the one collective (lws_amd/dist.py)
        # the true consistency 20 log10(|S| / |STFT(iSTFT(S)) - S|) (lws.pyx:140-144) of the same result, on the device
        # (lws_stft.hip), summed over all spectrograms of all ranks with the same all-reduce
        t0 = time.perf_counter()
        sums = lws_amd._capi.consistency_dev(state.data_ptr(), B, T, p.fsize, p.fshift, p.awin, p.swin, p.perfectrec,
                                              device=local_rank, stream=stream)
        extra["consistency_ms"] = 1e3 * (time.perf_counter() - t0)
        # (consistency_dev returns [|S|^2, |error|^2] per spectrogram: the pair reversed is what reduce_residual sums)
        _e2, _p2, extra["consistency_db_after"] = reduce_residual(sums[:, ::-1], device=coll_dev)
    del mags, state
    torch.cuda.empty_cache()

    if args.no_extras:
        wanted = []
    elif args.extras is not None:
        wanted = [x for x in args.extras.split(",") if x]
    elif world == 1 and args.config == "2":
        wanted = ["2-default", "2-single", "2-T1024", "2-q2", "2-q8", "2-q3", "2-frac", "2-speech", "2-q5", "2-q8w", "2-q16", "2-l8", "2-f501", "2-f257", "2-fp64", "4shard", "3", "3-b1024", "3-q16", "3-fp64", "host_api", "1", "5", "5-f16"]
    else:
        wanted = ["4shard"] if args.config == "2" else []
    if args.no_default_schedule and "2-default" in wanted:
        wanted.remove("2-default")
    cfgs = {}
    for name in wanted:
        try:
            if name == "2-default":
                # the reference's default schedule on the headline shape (51.8 effective sweeps; the first ~38 are no-ops)
                blk, keep = run_batch_config("2", max(1, args.steps), 1, schedule="default")
                extra["default_schedule"] = blk
                del keep
            elif name == "2-single":
                # latency of ONE spectrogram of the headline shape: its passes over HBM are pipelined over several workgroups
                blk, keep = run_batch_config("2", 2, 1, B=1, checks=False)
                extra["single_spectrogram"] = {"wall_ms": blk["ms_per_step"], "kernel_ms": blk["roofline"]["kernel_ms_per_step"],
                                               "kernel": blk["roofline"]["kernel"]}
                del keep
            elif name == "3":
                cfgs["3"] = run_config3(torch, lws_amd, dev, local_rank, rank, stream, sync_all, args.force_generic)
            elif name == "3-b1024":
                # config 3's pipeline on 1024 spectrograms: the no-future and online stages run one workgroup per spectrogram, two of
                # them per CU side by side when the batch is larger than the chip
                cfgs["3-b1024"] = run_config3(torch, lws_amd, dev, local_rank, rank, stream, sync_all, args.force_generic, B=1024)
            elif name == "3-q16":
                # the same pipeline at hop = frame/16 (lws(1024, 64)): no LDS engine takes its no-future and online stages -- the
                # team engine (round 6; the generic engine: 145 ms and 30.8 s for 64 spectrograms), the batch stage on the band engine
                cfgs["3-q16"] = run_config3(torch, lws_amd, dev, local_rank, rank, stream, sync_all, args.force_generic, fsize=1024, fshift=64, tag="3-q16")
            elif name == "host_api":
                cfgs["host_api"] = run_host_api(torch, lws_amd, dev, local_rank)
            elif name == "1":
                cfgs["1"] = run_config1(lws_amd, local_rank)
            elif name == "2-fp64":
                cfgs["2-fp64"] = run_fp64(torch, lws_amd, dev, local_rank)
            elif name == "3-fp64":
                cfgs["3-fp64"] = run_config3_fp64(torch, lws_amd, dev, local_rank)
            elif name in BATCH_CONFIGS:
                if BATCH_CONFIGS[name]["storage"] == "fp16" and not have_f16:
                    cfgs[name] = {"skipped": "this build has no fp16 storage mode"}
                    continue
                big = name.startswith("5")
                blk, keep = run_batch_config(name, 1 if big else 2, 1)
                del keep
                cfgs[name] = blk
                if big:   # the schedule BASELINE names for config 5 (get_thresholds(200, 100, 0.1, 1)), beside the dense roofline
                    blk2, keep = run_batch_config(name, 1, 0, schedule="default", checks=False)
                    del keep
                    blk["default_schedule"] = {k: blk2[k] for k in ("ms_per_step", "value", "active_value", "effective_sweeps")}
                    blk["default_schedule"]["algorithmic_GBs"] = blk2["roofline"]["achieved"]
            else:
                cfgs[name] = {"skipped": "unknown config"}
        except Exception as e:   # a failing extra must not take the headline line with it
            cfgs[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
    if cfgs:
        extra["configs"] = cfgs

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:   # rank 0's host cores, outside every timed region
        cpu = cpu_baseline(p.W, min(T, 500), F, iters, plan=None if args.force_generic else p.plan())
    if world > 1:
        dist.barrier()

    if rank == 0:
        head["fshift"] = p.fshift
        full = {"headline": {k: v for k, v in head.items() if k != "roofline"}, "roofline": roof, "cpu_baseline": cpu, "extra": extra,
                "parity_of_timed_workload": ("measured in this run (parity_measured_in_this_run: the timed plan against the CPU reference on one "
                                             "spectrogram from a random-phase start); the timed start (zero phases, run_lws(np.abs(X))) is "
                                             "ill-conditioned (DESIGN 6): fp32 is held to magnitudes 1e-6 + consistency 0.05 dB there, the "
                                             "schedule by the fp64 plan; the default schedule (extra.default_schedule) is checked value by value "
                                             "against the oracle in tests/test_gpu_parity.py")}
        extra_file = None
        try:   # everything that is not the contract's line: a file next to the run (gpurun_out/ travels back from the GPU box)
            d = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else ROOT
            extra_file = args.extra_file or os.path.join(d, "bench_extra.json")
            with open(extra_file, "w") as f:
                json.dump(_clean(full), f, indent=1, allow_nan=False)
            extra_file = os.path.relpath(extra_file, ROOT)
        except Exception:
            extra_file = None
        for l in summary_lines(extra):
            print(l)
        # parity of the timed arithmetic path, measured in this run (cpu_baseline leg: the oracle is the checker there, outside every
        # timed region); the timed start itself (zero phases) is ill-conditioned and is held to its properties
        par = (cpu or {}).pop("parity", None) if cpu else None
        if par:
            full["parity_measured_in_this_run"] = par
            ra, oe = par.get("fp64_reassociation") or {}, par.get("order_exact_fp32") or {}
            # (a compact copy for the line; `tail` = bins off by more than 1e-2 of mean|S| -- near-cancelling sums -- and, beside the timed
            #  kernel's figures, the order-exact fp32 engine's on the same input: what fp32 state costs whatever the kernel)
            also_parity = {"rel_l2": par["rel_l2"], "median": par["median_over_mean"], "p999": par["p999_over_mean"], "tail": par["bins_off_by_1e-2_mean"],
                           "bins": par["bins"], "order_exact_fp32_rel_l2": oe.get("rel_l2"), "order_exact_fp32_tail": oe.get("bins_off_by_1e-2_mean"),
                           "fp64_ref_vs_itself_rel_l2": ra.get("rel_l2")}
            notes = ("parity measured in this run (also.parity): 3 x %dx%d, %d dense sweeps from random phases, timed plan vs %s, in units of mean|S|; "
                     "bars 1e-3 / 1e-6 / 1e-3" % (min(T, 500), F, iters, "oracle/_ref" if cpu.get("kind") == "reference" else "oracle"))
        else:
            notes = "parity of this arithmetic path: tests/test_gpu_parity.py (not measured in this run: no CPU leg)"
        also = {}
        try:
            if par:
                also["parity"] = {k: (_sig(v, 3) if isinstance(v, float) else v) for k, v in also_parity.items()}
            c = extra.get("configs") or {}
            if "systolic" in (c.get("2-fp64") or {}):
                also["fp64_ms"] = float(c["2-fp64"]["systolic"]["kernel_ms"])
                also["fp64_kernel"] = c["2-fp64"]["systolic"]["kernel"]
                also["fp64_generic_ms"] = float(c["2-fp64"]["generic"]["kernel_ms"])
            if "total_wall_ms" in (c.get("3") or {}):
                also["config3_ms"] = float(c["3"]["total_wall_ms"])
            if "total_wall_ms" in (c.get("3-fp64") or {}):
                also["fp64_config3_ms"] = float(c["3-fp64"]["total_wall_ms"])
            if "online" in (c.get("3-q16") or {}):                       # lws(1024,64) music mode: online stage on the team engine (generic: 30.8 s / 64)
                also["q16_online_ms"] = float(c["3-q16"]["online"]["kernel_ms"])
            if "wall_ms" in (c.get("host_api") or {}):
                also["host_api_ms"] = float(c["host_api"]["wall_ms"])
            if "kernel_ps_per_bin_sweep" in (c.get("2-q8w") or {}):      # lws(2048,256) on the band engine (generic engine: 235)
                also["q8w_ps_per_bin_sweep"] = float(c["2-q8w"]["kernel_ps_per_bin_sweep"])
        except Exception:
            also = {}
        print(final_line(head, world, args.steps, args.warmup, p.fsize, roof, cpu, extra_file, notes, also))
        sys.stdout.flush()
    if world > 1:
        dist.destroy_process_group()


def run_config3(torch, lws_amd, dev, local_rank, rank, stream, sync_all, force_generic, B=256, T=500, fsize=1024, fshift=256, tag=None):
    """BASELINE config 3: the run_lws pipeline of lws(1024, 256, mode='music') -- 1 no-future sweep (W_ai, alpha 1),
    10 online iterations with look-ahead 3, 100 batch sweeps of the default schedule -- stage by stage on the device,
    each stage with its own roofline block (algorithmic bytes of SURVEY 8(d): 20 B per active bin-sweep, 4 B per
    inactive one; the online stage runs 1 + iters*(LA+1) frame sweeps per frame)."""
    F = fsize // 2 + 1
    pm = lws_amd.lws(fsize, fshift, mode="music", device=local_rank, force_generic=force_generic)
    planm = pm.plan()
    mags = torch.from_numpy(synth_magnitudes(B, T, F, 20260928 + rank * B)).to(dev)
    state = torch.empty((B, T, F), dtype=torch.complex64, device=dev)
    thr_nf = lws_amd.get_thresholds(pm.nofuture_iterations, pm.nofuture_alpha, pm.nofuture_beta, pm.nofuture_gamma)
    thr_on = lws_amd.get_thresholds(pm.online_iterations, pm.online_alpha, pm.online_beta, pm.online_gamma)
    thr_b = lws_amd.get_thresholds(pm.batch_iterations, pm.batch_alpha, pm.batch_beta, pm.batch_gamma)
    stages = [
        ("nofuture", thr_nf, 1, lambda: planm.nofuture_dev(state.data_ptr(), B, T, thr_nf, wsel=1, stream=stream)),
        ("online", thr_on, pm.look_ahead + 1, lambda: planm.online_dev(state.data_ptr(), B, T, thr_on, pm.look_ahead, fsize / fshift, stream=stream)),
        ("batch", thr_b, 1, lambda: planm.batch_dev(state.data_ptr(), B, T, thr_b, stream=stream)),
    ]
    c3 = {"workload": "%srun_lws(mode='music') on %d spectrograms of %d x %d, lws(%d,%d): 1 no-future sweep, "
                      "10 online iterations (look-ahead 3), 100 default-schedule batch sweeps" % ("BASELINE config 3: " if tag is None else "", B, T, F, fsize, fshift)}
    # bin-sweeps of each stage (what its thresholds let through), from the state each stage starts from: an untimed pass
    work = {}
    state.copy_(mags)
    for name, thr, sweeps_per_thr, fn in stages:
        cur = state.abs()
        mean = cur.mean(dim=(1, 2), keepdim=True)
        act = sum(float((cur > float(t) * mean).sum().item()) for t in thr) * sweeps_per_thr
        nominal = float(B) * T * F * len(thr) * sweeps_per_thr
        if name == "online":   # the online driver also runs one initial sweep (threshold 0) per frame
            act += float(B) * T * F
            nominal += float(B) * T * F
        work[name] = (act, nominal)
        fn()
    for rep in range(2):   # second repetition is the one reported
        state.copy_(mags)
        sync_all()
        t_all = time.perf_counter()
        for name, thr, sweeps_per_thr, fn in stages:
            t0 = time.perf_counter()
            fn()
            info = planm.last_kernel()
            torch.cuda.synchronize()
            wall = 1e3 * (time.perf_counter() - t0)
            act, nominal = work[name]
            alg = 16.0 * act + 4.0 * nominal
            traffic, tsrc = load_traffic(info["name"], tag or ("3" if B == 256 else "3-b%d" % B), stage=name)
            Wst = pm.W_ai if name == "nofuture" else pm.W      # (the online stage mixes W, W_ai, W_af: priced with W)
            c3[name] = {"wall_ms": wall, "kernel_ms": info["ms"], "kernel": info["name"], "bin_sweeps": nominal, "active_bin_sweeps": act,
                        # batch: vector-ALU issue; no-future / online: the dependent chain of a step (barrier rounds, LDS round
                        # trips), neither HBM nor arithmetic throughput (profiles/r03_pmc_sq_online.json)
                        "roofline": roofline_block(alg, info["ms"], act, Wst, traffic, tsrc, info["name"], 1.0, "valu" if name == "batch" else "latency")}
        c3["total_wall_ms"] = 1e3 * (time.perf_counter() - t_all)
    c3["iterations"] = {"nofuture": pm.nofuture_iterations, "online": pm.online_iterations, "batch": pm.batch_iterations,
                        "look_ahead": pm.look_ahead}
    c0 = lws_amd._capi.consistency_dev(mags[:4].to(torch.complex64).data_ptr(), 4, T, fsize, fshift, pm.awin, pm.swin, pm.perfectrec,
                                       device=local_rank, stream=stream).sum(axis=0)
    c1 = lws_amd._capi.consistency_dev(state[:4].data_ptr(), 4, T, fsize, fshift, pm.awin, pm.swin, pm.perfectrec,
                                       device=local_rank, stream=stream).sum(axis=0)
    c3["checks"] = {"max_rel_magnitude_error": float(((state.abs() - mags).abs().max() / mags.max()).item()),
                    "consistency_db_before": float(10 * np.log10(c0[0] / c0[1])), "consistency_db_after": float(10 * np.log10(c1[0] / c1[1]))}
    return c3


def run_host_api(torch, lws_amd, dev, local_rank, B=256, T=500, iters=100):
    """BASELINE config 2 through the host-array entry point (lws.pyx:209-258 / python/README.md:96-100 hand numpy arrays in and
    get numpy arrays back): plan.batch(complex128 (B,T,F)) -> complex128, wall time of the call.  The library cuts the batch into
    chunks and overlaps narrowing to complex64 / H2D / sweeps / D2H / widening (lws_capi.hip: run_host_pipelined); the floor is
    the kernel time plus the first upload and the last download, which nothing overlaps."""
    F = 513
    p = lws_amd.lws(1024, 256, device=local_rank)
    plan = p.plan()
    M = synth_magnitudes(B, T, F, 20260928).astype(np.complex128)
    thr = np.zeros(iters)
    walls, keep = [], []
    for rep in range(4):
        t0 = time.perf_counter()
        keep.append(plan.batch(M, thr))
        walls.append(1e3 * (time.perf_counter() - t0))
        if len(keep) > 2:
            keep.pop(0)          # (freed outside the timed call: returning 1 GB to the OS costs 20-40 ms by itself)
    out = keep[-1]
    # the same volume with complex input (random phases): goes up as 8 bytes per bin instead of 4
    Mc = (M * np.exp(2j * np.pi * np.random.default_rng(1).random(M.shape))).astype(np.complex128)
    walls_c = []
    for rep in range(3):
        t0 = time.perf_counter()
        keep.append(plan.batch(Mc, thr))
        walls_c.append(1e3 * (time.perf_counter() - t0))
        keep.pop(0)
    del Mc
    # pinned copy rate of this box, both directions, for the "transfer time" the wall is compared with
    h = torch.empty(1 << 28, dtype=torch.uint8).pin_memory()
    g = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    rates = []
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.copy_(h, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
        h.copy_(g, non_blocking=True); torch.cuda.synchronize(); t2 = time.perf_counter()
        rates = [(1 << 28) / (t1 - t0) / 1e9, (1 << 28) / (t2 - t1) / 1e9]
    d = torch.from_numpy(M.astype(np.complex64)).to(dev)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.batch_dev(d.data_ptr(), B, T, thr); torch.cuda.synchronize()
        dev_ms = 1e3 * (time.perf_counter() - t0)
    batch_checks = {"max_rel_magnitude_error": float(np.abs(np.abs(out) - np.abs(M)).max() / np.abs(M).max()), "finite": bool(np.isfinite(out).all())}
    bytes_c64 = float(B) * T * F * 8
    xfer_ms = 1e3 * bytes_c64 / (min(rates) * 1e9)
    del d, h, g, keep, out
    # BASELINE config 3 the same way: run_lws(mode='music') on numpy magnitudes (no-future -> online -> batch: the first two run one
    # workgroup per spectrogram, so the call is one chunk of 256 and only the host passes overlap with the copies)
    pm = lws_amd.lws(1024, 256, mode="music", device=local_rank)
    walls3, keep3 = [], []
    for rep in range(3):
        t0 = time.perf_counter()
        keep3.append(pm.run_lws(M))
        walls3.append(1e3 * (time.perf_counter() - t0))
        if len(keep3) > 1:
            keep3.pop(0)
    out = keep3[-1]
    music = {"workload": "BASELINE config 3 through lws(1024,256,mode='music').run_lws(numpy complex128 %dx%dx%d) -> complex128" % (B, T, F),
             "wall_ms": min(walls3[1:]), "wall_ms_all_calls": walls3,
             "checks": {"max_rel_magnitude_error": float(np.abs(np.abs(out) - np.abs(M)).max() / np.abs(M).max()), "finite": bool(np.isfinite(out).all())}}
    out = keep3 = None
    # the same entry point on a quarter and on four times the batch (what the chunking costs a small call, and whether a large one
    # reaches the device-resident rate): wall of the call / spectrogram, against config 2's
    other = {}
    if B == 256:
        for B2 in (64, 1024):
            M2 = synth_magnitudes(B2, T, F, 20260928).astype(np.complex128)
            w2, keep2 = [], []
            for rep in range(3):
                t0 = time.perf_counter()
                keep2.append(plan.batch(M2, thr))
                w2.append(1e3 * (time.perf_counter() - t0))
                if len(keep2) > 1:
                    keep2.pop(0)
            d2 = torch.from_numpy(M2.astype(np.complex64)).to(dev)
            for rep in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                plan.batch_dev(d2.data_ptr(), B2, T, thr); torch.cuda.synchronize()
                dev2 = 1e3 * (time.perf_counter() - t0)
            other["B%d" % B2] = {"wall_ms": min(w2[1:]), "device_resident_ms": dev2, "ratio": min(w2[1:]) / dev2,
                                 "value": float(B2) * T * F * iters / (min(w2[1:]) * 1e-3)}
            del M2, keep2, d2
    return {"run_lws_music": music, "other_batch_sizes": other, "workload": "BASELINE config 2 through plan.batch(numpy complex128 %dx%dx%d) -> complex128, %d dense sweeps" % (B, T, F, iters),
            "wall_ms": min(walls[1:]), "wall_ms_all_calls": walls, "first_call_includes": "pinned staging buffers (hipHostMalloc) and scratch",
            "input": "real-valued magnitudes (the documented usage run_lws(np.abs(X))): 4 B/bin over the bus on the way up, 8 on the way down",
            "complex_input": {"wall_ms": min(walls_c[1:]), "wall_ms_all_calls": walls_c, "note": "random phases: 8 B/bin each way"},
            "value": float(B) * T * F * iters / (min(walls[1:]) * 1e-3), "device_resident_ms": dev_ms,
            "pinned_copy_GBs": {"h2d": rates[0], "d2h": rates[1]}, "bytes_over_the_bus_each_way": bytes_c64,
            "transfer_ms_each_way_at_pinned_rate": xfer_ms, "wall_over_max_transfer_kernel": min(walls[1:]) / max(xfer_ms, dev_ms),
            "checks": batch_checks}


def run_config3_fp64(torch, lws_amd, dev, local_rank, B=256, T=500):
    """BASELINE config 3 in the reference's own arithmetic type: run_lws(mode='music') of an fp64 plan on device-resident complex128
    spectrograms, stage by stage (kernel names say which engine served each: the LDS engines of the no-future / online stages and the
    fp64 systolic engine, or the order-exact generic engine)."""
    F = 513
    pm = lws_amd.lws(1024, 256, mode="music", precision="fp64", device=local_rank)
    plan = pm.plan()
    M = synth_magnitudes(B, T, F, 20260928).astype(np.complex128)
    thr = [lws_amd.get_thresholds(pm.nofuture_iterations, pm.nofuture_alpha, pm.nofuture_beta, pm.nofuture_gamma),
           lws_amd.get_thresholds(pm.online_iterations, pm.online_alpha, pm.online_beta, pm.online_gamma),
           lws_amd.get_thresholds(pm.batch_iterations, pm.batch_alpha, pm.batch_beta, pm.batch_gamma)]
    d = torch.from_numpy(M).to(dev)
    stream = torch.cuda.current_stream().cuda_stream
    stages = [("nofuture", lambda: plan.nofuture_dev(d.data_ptr(), B, T, thr[0], wsel=1, stream=stream)),
              ("online", lambda: plan.online_dev(d.data_ptr(), B, T, thr[1], pm.look_ahead, 4.0, stream=stream)),
              ("batch", lambda: plan.batch_dev(d.data_ptr(), B, T, thr[2], stream=stream))]
    out = {"workload": "BASELINE config 3 in fp64: run_lws(mode='music', precision='fp64') on %d spectrograms of %d x %d, complex128 resident in HBM" % (B, T, F)}
    for rep in range(2):
        d.copy_(torch.from_numpy(M))
        torch.cuda.synchronize(); t_all = time.perf_counter()
        for name, fn in stages:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            k = plan.last_kernel()
            out[name] = {"wall_ms": 1e3 * (time.perf_counter() - t0), "kernel_ms": k["ms"], "kernel": k["name"]}
        out["total_wall_ms"] = 1e3 * (time.perf_counter() - t_all)
    res = d.cpu().numpy()
    out["checks"] = {"max_rel_magnitude_error": float(np.abs(np.abs(res) - np.abs(M)).max() / np.abs(M).max()), "finite": bool(np.isfinite(res).all())}
    plan.close()
    return out


def run_fp64(torch, lws_amd, dev, local_rank, B=256, T=500, iters=100):
    """BASELINE config 2's volume in the reference's own arithmetic type (lwslib.h:6-26 is double throughout): an fp64 plan on
    device-resident complex128 spectrograms.  Its batch sweeps run on the fp64 systolic engine (lws_sys64.hip); the order-exact
    generic engine (LWS_FORCE_GENERIC) is timed beside it."""
    from lws_amd import _capi
    F = 513
    p = lws_amd.lws(1024, 256, device=local_rank)
    M = synth_magnitudes(B, T, F, 20260928).astype(np.complex128)
    thr = np.zeros(iters)
    n = float(B) * T * F * iters
    out = {"workload": "%d spectrograms x %d frames x %d bins, lws(1024,256), %d dense batch-LWS sweeps, fp64, complex128 resident in HBM" % (B, T, F, iters),
           "algorithmic_bytes_per_bin_sweep": 40}
    for key, generic in (("systolic", False), ("generic", True)):
        plan = _capi.Plan(F, p.W, precision="fp64", force_generic=generic, device=local_rank)
        d = torch.from_numpy(M).to(dev)
        ms, wall = [], []
        for rep in range(3 if not generic else 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            plan.batch_dev(d.data_ptr(), B, T, thr); torch.cuda.synchronize()
            wall.append(1e3 * (time.perf_counter() - t0)); ms.append(plan.last_kernel()["ms"])
        k = plan.last_kernel()
        res = d.cpu().numpy() if not generic else None
        out[key] = {"kernel": k["name"], "kernel_ms": min(ms[1:]), "wall_ms": min(wall[1:]), "launches": k.get("launches"),
                    "value": n / (min(wall[1:]) * 1e-3), "algorithmic_GBs": 40.0 * n / (min(ms[1:]) * 1e-3) / 1e9}
        if res is not None:
            out[key]["checks"] = {"max_rel_magnitude_error": float(np.abs(np.abs(res) - np.abs(M)).max() / np.abs(M).max()), "finite": bool(np.isfinite(res).all())}
        plan.close()
        del d
    out["speedup_over_generic"] = out["generic"]["kernel_ms"] / out["systolic"]["kernel_ms"]
    return out


def run_config1(lws_amd, local_rank):
    """BASELINE config 1: a single 5 s 16 kHz mono clip, 512-point STFT hop 128 (628 x 257), batch LWS through the reference's
    own Python entry point lws.lws(512,128).run_lws.  Taken literally (10 iterations, default alpha = 100) no bin is ever
    above a threshold and the call is a no-op, so the block also times the schedule SURVEY 8(d)
    names for this config, get_thresholds(10, 1, 0.1, 1).  Wall time of the whole call from numpy to numpy, the first one
    including plan creation; beside it the reference CPU path (oracle/_ref when it travelled, else the fp64 port) on the same clip."""
    from oracle.oracle import Oracle, RefLib
    x = np.random.default_rng(0).standard_normal(80000)
    out = {"workload": "BASELINE config 1: 5 s of 16 kHz noise, lws(512,128): 628 x 257, 10 batch-LWS iterations via run_lws (numpy in, numpy out)"}
    for label, kw in (("literal_defaults", dict(batch_iterations=10)), ("alpha_1", dict(batch_iterations=10, batch_alpha=1.0))):
        t0 = time.perf_counter()
        p = lws_amd.lws(512, 128, device=local_rank, **kw)
        X = p.stft(x)
        M = np.abs(X)
        t1 = time.perf_counter()
        Y = p.run_lws(M)                 # first call: creates the device plan
        t2 = time.perf_counter()
        walls = []
        for rep in range(5):
            t3 = time.perf_counter()
            Y = p.run_lws(M)
            walls.append(1e3 * (time.perf_counter() - t3))
        thr = lws_amd.get_thresholds(10, kw.get("batch_alpha", 100), 0.1, 1)
        orc = Oracle()
        ref = orc.batch_lws(M.astype(np.complex128), p.W, thr)           # the checker
        cpu_kind = "port (oracle/lws_oracle.c, fp64, 1 thread)"
        tc = time.perf_counter()
        if RefLib.available():   # the reference's own LWSQ4 (lwslib.cpp:153-280), compiled in place: the 10 sweeps of batch_lws
            import ctypes as C
            from oracle.oracle import split_weights
            rl = RefLib()
            wr, wi, wf = split_weights(p.W)
            er, ei = orc.extend(M.astype(np.complex128), p.L, int(p.Q))
            amp = np.ascontiguousarray(np.abs(er + 1j * ei))
            mean = float(M.mean())
            tc = time.perf_counter()
            for t_i in thr:
                rl.fn["LWSQ4"](C.c_void_p(er.ctypes.data), C.c_void_p(ei.ctypes.data), C.c_void_p(wr.ctypes.data), C.c_void_p(wi.ctypes.data),
                               C.c_void_p(wf.ctypes.data), C.c_void_p(amp.ctypes.data), int(M.shape[1]), int(M.shape[0]), int(p.L), float(t_i * mean))
            cpu_kind = "reference (oracle/_ref: lwslib.cpp LWSQ4, fp64, 1 thread; the sweeps only, without the wrapper's preparation)"
        else:
            orc.batch_lws(M.astype(np.complex128), p.W, thr)
        cpu_ms = 1e3 * (time.perf_counter() - tc)
        err = np.abs(Y - ref)
        out[label] = {"frames": int(M.shape[0]), "bins": int(M.shape[1]), "first_call_ms_incl_plan_creation": 1e3 * (t2 - t1),
                      "wall_ms": min(walls), "value": float(M.size) * 10 / (min(walls) * 1e-3),
                      "kernel": p.plan().last_kernel()["name"],
                      "cpu_reference_ms": cpu_ms, "cpu_kind": cpu_kind,
                      "updated_bins_fraction": float(np.mean(Y != M)),
                      "checks": {"rel_l2_vs_cpu": float(np.linalg.norm(err) / max(np.linalg.norm(ref), 1e-300)),
                                 "max_rel_magnitude_error": float(np.abs(np.abs(Y) - M).max() / M.max())}}
    return out


if __name__ == "__main__":
    main()
