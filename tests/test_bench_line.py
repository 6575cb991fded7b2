"""CPU: the ONE line bench.py prints for the driver stays compact and strict JSON (round 3's 24 KB line came back unparsed),
and config 4's 8-way split is what lws_amd/dist.py produces."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (torch is only imported inside bench.main)


def _canned(world):
    W = np.zeros((4, 4, 6), dtype=np.complex128)
    W[:, :, :] = 0.1 + 0.2j
    head = {"value": 1.970123456789e11 * world, "ms_per_step": 33.312345678, "storage": "fp32", "batch_per_gpu": 256, "frames": 500,
            "bins": 513, "iters": 100, "schedule": "dense", "fshift": 256,
            "workload": "x" * 400, "data": "y" * 200}
    roof = bench.roofline_block(131.328e9, 32.4123456, 6.5664e9, W, 2.034e10, "profiles/r04_pmc_traffic.json (" + "z" * 300 + ")",
                                "systolic_q4_l5_hann_with_a_rather_long_kernel_name", 1.0, "valu")
    roof["frac_of_measured_copy"] = 0.654321
    cpu = {"value": 3.68e7, "unit": "bin*iter/s", "cores": 1, "kind": "reference", "sample": "1 spectrogram 500x513, 23 dense sweeps, fp64, single thread",
           "all_cores_value": 5.5e8, "all_cores": 16, "hw_threads": 256, "physical_cores": 128, "cpu_quota": 16.0}
    return head, roof, cpu


def test_final_line_is_compact_strict_json():
    for world in (1, 2, 4, 8):
        head, roof, cpu = _canned(world)
        txt = bench.final_line(head, world, 20, 5, 1024, roof, cpu, "gpurun_out/bench_extra.json", "n" * 500,
                               {"fp64_ms": 97.123456, "fp64_kernel": "systolic_fp64_q4", "fp64_generic_ms": 836.2, "config3_ms": 73.1, "host_api_ms": 49.1})
        assert len(txt) < 2048 and "\n" not in txt
        d = json.loads(txt, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))     # NaN / Infinity would raise
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, k
        assert d["n_gpus"] == world and d["config"]["parallelism"] == "shard%d" % world and "workload" in d["config"]
        r = d["roofline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms_per_step", "algorithmic_bytes_per_launch"):
            assert k in r, k
        assert math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-4)
        assert math.isclose(r["achieved"], 131.328e9 / 32.4123456e-3 / 1e9, rel_tol=1e-4)
        assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "reference"


def test_the_line_keeps_its_measured_parity():
    """Round 6: the parity of the timed arithmetic path is measured in the run (cpu_baseline leg) and travels in `also.parity`;
    with realistic field sizes nothing is dropped to stay under the limit."""
    head, roof, cpu = _canned(1)
    also = {"parity": {"rel_l2": 1.26e-3, "median": 2.0e-7, "p999": 3.6e-4, "tail": 57, "bins": 769500, "order_exact_fp32_rel_l2": 4.1e-4, "order_exact_fp32_tail": 41, "fp64_ref_vs_itself_rel_l2": 1.7e-12},
            "fp64_ms": 97.123456, "fp64_kernel": "systolic_fp64_q4", "fp64_generic_ms": 836.2, "config3_ms": 73.1, "fp64_config3_ms": 402.3, "host_api_ms": 49.1,
            "q8w_ps_per_bin_sweep": 43.4}
    notes = ("parity measured in this run (also.parity): 3 x 500x513, 100 dense sweeps from random phases, timed plan vs oracle/_ref, in units of mean|S|; "
             "bars 1e-3 / 1e-6 / 1e-3")
    txt = bench.final_line(head, 1, 20, 5, 1024, roof, cpu, "gpurun_out/bench_extra.json", notes, also)
    d = json.loads(txt)
    assert len(txt) < bench.MAX_LINE and d["notes"] == notes and d["also"]["parity"]["tail"] == 57 and d["cpu_baseline"]["cpu_quota"] == 16.0
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["roofline_axis"] == "hbm" and d["roofline"]["limiter"] == "valu"


def test_cpu_baseline_times_only_the_sweeps_of_its_threads():
    """bench.cpu_baseline on this box's cores (a short budget): the all-cores figure -- one spectrogram per physical core, inputs
    built before the clock starts, the threads released together -- must scale with the cores (round 5 started its clock before the
    threads built their inputs under the GIL: 8.5x on 256 threads)."""
    import lws_amd
    # (a timed region of a few hundred ms per thread: threads woken together start on one core and take a while to spread)
    # (a shared build container is noisy -- other test workers, other tenants: the best of up to three measurements is judged)
    for attempt in range(3):
        cpu = bench.cpu_baseline(lws_amd.lws(256, 64).W, 120, 129, 600, budget_s=1.0)
        if cpu["all_cores"] < 4 or cpu["all_cores_value"] >= 0.4 * cpu["all_cores"] * cpu["value"]:
            break
    assert cpu["cores"] == 1 and cpu["all_cores"] >= 1 and cpu["hw_threads"] >= cpu["all_cores"] and "parity" not in cpu
    assert cpu["value"] > 1e6
    if cpu["all_cores"] >= 4:      # (a shared build container is noisy: the GPU box's figure is the one that is quoted)
        assert cpu["all_cores_value"] >= 0.4 * cpu["all_cores"] * cpu["value"], cpu
    if cpu["all_cores"] >= 8:
        assert cpu["all_cores_value"] >= 4 * cpu["value"], cpu


def test_non_finite_numbers_become_null():
    head, roof, cpu = _canned(1)
    roof["traffic"] = float("nan")
    roof["hbm_measured_frac"] = float("inf")
    txt = bench.final_line(head, 1, 3, 1, 1024, roof, None)
    d = json.loads(txt)
    assert d["roofline"]["traffic"] is None and d["roofline"]["hbm_measured_frac"] is None and d["cpu_baseline"] is None
    assert "NaN" not in txt and "Infinity" not in txt


def test_summary_lines_are_short_and_never_json_objects():
    extra = {"configs": {"5": {"ms_per_step": 4500.0, "roofline": {"frac": 0.41, "valu": {"frac_naive": 0.5}, "kernel_ms_per_step": 4480.0, "kernel": "k" * 500}},
                         "bad": {"error": "e" * 1000}, "odd": {"roofline": {}}}}
    ls = bench.summary_lines(extra)
    assert ls and all(len(l) <= 300 and l.startswith("#") for l in ls)


def test_config4_split_is_eight_contiguous_shards_of_1024():
    from lws_amd.dist import shard_range
    got = [shard_range(8192, r, 8) for r in range(8)]
    assert got == [(1024 * r, 1024 * (r + 1)) for r in range(8)]
    # ragged totals: contiguous, complete, sizes differ by at most one
    for n, w in ((8191, 8), (5, 8), (1000, 3)):
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in rs]
        assert max(sizes) - min(sizes) <= 1


def _launch(tmp_path, n, argv, have, body, backend=None):
    """bench.self_launch around a stand-in rank script (no GPU here): returns (rc, what the ranks wrote)."""
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, json\nout = os.environ['OUT_DIR']\n"
                      "json.dump({k: os.environ.get(k) for k in ('RANK','LOCAL_RANK','WORLD_SIZE','LOCAL_WORLD_SIZE','MASTER_ADDR','MASTER_PORT')} | {'argv': sys.argv[1:]},"
                      " open(os.path.join(out, 'r%s.json' % os.environ.get('RANK', 'solo')), 'w'))\n" + body)
    code = ("import sys, os; sys.path.insert(0, %r); import bench; os.environ['OUT_DIR'] = %r; %s"
            "raise SystemExit(bench.self_launch(%d, %r, script=%r, have=%d))"
            % (ROOT, str(tmp_path), ("os.environ['LWS_BENCH_BACKEND'] = %r; " % backend) if backend else "", n, argv, str(script), have))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LWS_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    got = {f: json.load(open(tmp_path / f)) for f in sorted(os.listdir(tmp_path)) if f.endswith(".json")}
    return out, got


def test_self_launch_starts_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus N` without a launcher (the way the driver starts N = 1) starts its own N ranks with the launcher's
    environment; rank 0's stdout is the launcher's."""
    out, got = _launch(tmp_path, 4, ["--gpus", "4", "--steps", "3", "--warmup=1"], 8, "if os.environ['RANK'] == '0': print('{\"line\": 1}')\nelse: print('noise')\n")
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == '{"line": 1}'                       # only rank 0 reaches stdout
    assert sorted(got) == ["r0.json", "r1.json", "r2.json", "r3.json"]
    ports = set()
    for r in range(4):
        g = got["r%d.json" % r]
        assert g["RANK"] == str(r) and g["LOCAL_RANK"] == str(r) and g["WORLD_SIZE"] == "4" and g["LOCAL_WORLD_SIZE"] == "4"
        assert g["MASTER_ADDR"] == "127.0.0.1"
        assert g["argv"] == ["--gpus", "4", "--steps", "3", "--warmup=1"]
        ports.add(g["MASTER_PORT"])
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_self_launch_runs_on_the_gpus_there_are(tmp_path):
    """RCCL needs a GPU per rank: --gpus 8 on a 2-GPU box runs 2 ranks (and says so); with one GPU the single rank runs without a
    process group; the gloo test backend lets ranks share a GPU."""
    out, got = _launch(tmp_path, 8, ["--steps", "2", "--gpus=8"], 2, "")
    assert out.returncode == 0 and "2 GPU(s) visible" in out.stderr
    assert sorted(got) == ["r0.json", "r1.json"] and got["r1.json"]["WORLD_SIZE"] == "2" and got["r0.json"]["argv"] == ["--gpus", "2", "--steps", "2"]
    for f in got:
        os.remove(tmp_path / f)
    out, got = _launch(tmp_path, 2, ["--gpus", "2"], 1, "")
    assert out.returncode == 0 and list(got) == ["rsolo.json"] and got["rsolo.json"]["WORLD_SIZE"] is None and got["rsolo.json"]["argv"] == ["--gpus", "1"]
    os.remove(tmp_path / "rsolo.json")
    out, got = _launch(tmp_path, 2, ["--gpus", "2"], 1, "", backend="gloo")
    assert out.returncode == 0 and sorted(got) == ["r0.json", "r1.json"]
    out, got = _launch(tmp_path, 2, ["--gpus", "2"], 0, "")
    assert out.returncode == 2 and "no GPU" in out.stderr


def test_self_launch_stops_the_job_when_a_rank_fails(tmp_path):
    import time as _t
    t0 = _t.time()
    out, got = _launch(tmp_path, 3, ["--gpus", "3"], 3, "import time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)\n")
    assert out.returncode == 7 and "rank 1 exited with code 7" in out.stderr
    assert _t.time() - t0 < 30                                          # the other ranks were stopped, not waited for


def test_traffic_profile_of_another_kernel_is_refused(tmp_path, monkeypatch):
    """roofline.traffic comes from committed rocprofv3 --pmc passes, not from the timed run: an entry taken on another kernel than
    the one the plan just ran is refused (traffic null, the reason in traffic_source), a changed kernel source marks it STALE."""
    assert bench.traffic_entry_matches({"kernel": "void lws::(anonymous namespace)::k_systolic<4, 5, 1ul, false, false, 0>(...)"}, "systolic_q4_l5_hann")
    assert not bench.traffic_entry_matches({"kernel": "void lws::(anonymous namespace)::k_systolic<4, 5, 1ul>(...)"}, "generic_skew_fp32")
    assert not bench.traffic_entry_matches({"kernel": "k_online4<...>"}, "systolic_q4_l5_hann")
    assert bench.traffic_entry_matches({"engine_kernel": "online_lds_fp32", "kernel": "k_online4"}, "online_lds_fp32")
    assert not bench.traffic_entry_matches({"engine_kernel": "systolic_q4_l5_hann", "kernel": "k_systolic"}, "systolic_q4_l3_hann")
    root = tmp_path / "repo"
    (root / "profiles").mkdir(parents=True)
    (root / "lws_amd" / "csrc").mkdir(parents=True)
    (root / "lws_amd" / "csrc" / "lws_systolic.hip").write_text("v1")
    monkeypatch.setattr(bench, "ROOT", str(root))
    prof = {"_sources": bench._source_hashes(),
            "configs": {"2": {"engine_kernel": "systolic_q4_l5_hann", "kernel": "k_systolic<4>", "hbm_bytes_per_launch": 2.0e10}}}
    (root / "profiles" / "r05_pmc_traffic.json").write_text(json.dumps(prof))
    t, src = bench.load_traffic("systolic_q4_l5_hann", "2")
    assert t == 2.0e10 and "STALE" not in src
    t, src = bench.load_traffic("generic_skew_fp32", "2")
    assert t is None and "refused" in src and "generic_skew_fp32" in src
    (root / "lws_amd" / "csrc" / "lws_systolic.hip").write_text("v2")
    t, src = bench.load_traffic("systolic_q4_l5_hann", "2")
    assert t == 2.0e10 and src.startswith("STALE (lws_systolic.hip")
    # the committed profile still matches what the headline plan runs
    monkeypatch.undo()
    t, src = bench.load_traffic("systolic_q4_l5_hann", "2")
    assert t and t > 1e10
