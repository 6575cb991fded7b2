#!/usr/bin/env python3
"""Turn the rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into the summaries kept under profiles/.
usage: collect_profiles.py <prof_dir> <out_dir> <tag>     (run on the GPU box or on merged gpurun_out/)"""
import csv, glob, json, os, sqlite3, sys

prof, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)


def db(sub):
    f = glob.glob(os.path.join(prof, sub, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def tables(con):
    return [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]


def find(con, prefix):
    for t in tables(con):
        if t == prefix or t.startswith(prefix):
            return t
    return None


for sub, suffix, what in (("trace", "", "python bench.py --steps 3 --warmup 1 --no-cpu-baseline  (MI355X; the default bench command: headline config 2 + every extra config block"),
                          ("trace_headline", "_headline", "python bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline  (MI355X; the headline workload alone: BASELINE config 2, 256 x 500 x 513, 100 dense sweeps")):
    con = db(sub)
    if not con:
        continue
    kt = find(con, "kernels")
    rows = con.execute(f"select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from {kt} group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(os.path.join(out, f"{tag}_kernel_stats{suffix}.csv"), "w", newline="") as fh:
        fh.write('"# rocprofv3 --kernel-trace --stats -- %s; durations in microseconds)"\n' % what)
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_us", "average_us", "min_us", "max_us", "percent"])
        for n, c, s_, a, mn, mx in rows:
            w.writerow([n[:200], c, round(s_ / 1e3, 3), round(a / 1e3, 3), round(mn / 1e3, 3), round(mx / 1e3, 3), round(100 * s_ / tot, 3)])

SHAPES = {"2": (256, 500, 513, 100, 20.0), "2-T1024": (256, 1024, 513, 100, 20.0), "2-q2": (256, 500, 513, 100, 20.0), "2-q8": (256, 500, 513, 100, 20.0), "2-f501": (256, 500, 501, 100, 20.0), "2-f257": (512, 500, 257, 100, 20.0), "2-q3": (256, 500, 385, 100, 20.0), "2-frac": (256, 500, 513, 100, 20.0), "2-speech": (512, 500, 201, 100, 20.0), "2-q5": (256, 500, 501, 40, 20.0), "2-q8w": (256, 500, 1025, 20, 20.0), "2-q16": (256, 500, 513, 10, 20.0), "2-l8": (256, 500, 513, 40, 20.0), "4shard": (1024, 500, 513, 100, 20.0),
          "5": (64, 56250, 1025, 200, 20.0), "5-f16": (64, 56250, 1025, 200, 10.0)}
res = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE -- python bench.py --config <cfg> "
               "--no-extras --steps 1 --warmup 1 --no-cpu-baseline (config 3: --extras 3); counters are KiB per dispatch. "
               "MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports 1/2 of the "
               "bytes -> doubled here; WRITE_SIZE is exact.  hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) * 1024 of the LAST "
               "launch of the update kernel (the timed step; the warm-up launch of the big shapes runs 3 sweeps only); the band engine launches "
               "once per pass of its sweep slots: its entry is the sum over the launches of the timed step (launches_per_step).",
       "configs": {}}
import hashlib
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
res["_sources"] = {}
for fn in ("lws_systolic.hip", "lws_online.hip", "lws_nofuture.hip", "lws_common.h", "lws_band.hip", "lws_band_core.h"):   # what bench.py's load_traffic compares with
    try:
        res["_sources"][fn] = hashlib.sha1(open(os.path.join(root, "lws_amd", "csrc", fn), "rb").read()).hexdigest()
    except OSError:
        pass


def engine_names(cfg):
    """lws_last_kernel_name of the profiled run (the bench line / extra file tools/profile.sh kept): {stage: name}"""
    try:
        if cfg == "3":
            c3 = json.load(open(os.path.join(prof, "extra_3.json")))["extra"]["configs"]["3"]
            return {st: c3[st]["kernel"] for st in ("nofuture", "online", "batch") if st in c3}
        last = [l for l in open(os.path.join(prof, "pmc_line_%s.json" % cfg)).read().splitlines() if l.startswith("{")][-1]
        return {"batch": json.loads(last)["roofline"]["kernel"]}
    except Exception:
        return {}


for cfg in list(SHAPES) + ["3"]:
    per = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        c2 = db(f"pmc_{cn}_{cfg}")
        if not c2:
            continue
        ct = find(c2, "counters_collection")
        cols = [r[1] for r in c2.execute(f"pragma table_info({ct})")]
        order = "dispatch_id" if "dispatch_id" in cols else "rowid"
        for n, v in c2.execute(f"select kernel_name, value from {ct} where counter_name=? order by {order}", (cn,)):
            per.setdefault(n, {}).setdefault(cn, []).append(v)
    if not per:
        continue
    ent = {}
    for n, d in per.items():
        key = None
        if "k_systolic" in n: key = "batch"
        elif "k_band<" in n: key = "batch"
        elif "k_nofuture" in n: key = "nofuture"
        elif "k_online" in n: key = "online"
        if key is None or "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        f, w = d["FETCH_SIZE"][-1], d["WRITE_SIZE"][-1]
        per_step = 1
        if "k_band<" in n:   # one launch per pass of NS sweeps: the step (bench: --steps 1 --warmup 1) is the second half of the launches seen
            per_step = max(1, len(d["FETCH_SIZE"]) // 2)
            f, w = sum(d["FETCH_SIZE"][-per_step:]), sum(d["WRITE_SIZE"][-per_step:])
        ent[key] = {"launches_per_step": per_step, "engine_kernel": engine_names(cfg).get(key), "kernel": n[:160], "launches_seen": len(d["FETCH_SIZE"]), "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                    "hbm_bytes_per_launch": (2 * f + w) * 1024}
    if cfg in SHAPES and "batch" in ent:
        B, T, F, it, bpb = SHAPES[cfg]
        e = ent["batch"]
        e["algorithmic_bytes_per_launch"] = B * T * F * it * bpb
        e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
        res["configs"][cfg] = e
    elif cfg == "3":
        res["configs"]["3"] = ent
json.dump(res, open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res["configs"], indent=1)[:3000])
