#!/bin/bash
# rocprofv3 evidence for the fp64 systolic engine (profiles/<tag>_sys64_*): kernel-trace stats of config 2's volume on an fp64
# plan, SQ counters (three passes) and FETCH_SIZE / WRITE_SIZE (one pass each, as MI355X_MICROARCH.md prescribes).
# usage (GPU box, repo root):  bash tools/profile_sys64.sh r04            (config 2's volume, 64 frames in flight)
#                              bash tools/profile_sys64.sh r05 --wide-one (lws(2048,512), 256 x 250 x 1025, 40 sweeps: 128 frames in flight)
TAG=${1:-r04}
MODE=${2:---one}
SUF=$([ "$MODE" = "--one" ] && echo "" || echo "_wide")
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/prof_sys64_$TAG$SUF
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python tools/time_sys64.py $MODE > $OUT/time.out 2> $OUT/trace.err
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_$ctr -o $TAG -- python tools/time_sys64.py $MODE > /dev/null 2> $OUT/pmc_$ctr.err
done
python3 - $OUT $TAG$SUF $MODE <<'PY'
import glob, json, os, sqlite3, sys
out, tag, mode = sys.argv[1], sys.argv[2], sys.argv[3]
wide = mode != "--one"
res = {"_how": "tools/profile_sys64.sh: rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python tools/time_sys64.py " + mode + " "
               + ("(lws(2048,512): 256 x 250 x 1025, 40 dense sweeps, fp64 plan, 3 calls of 20 launches of 2 sweeps; 128 frames in flight, two waves per sweep slot)" if wide else
                  "(256 x 500 x 513, 100 dense sweeps, fp64 plan, 3 calls of 25 launches of 4 sweeps)") + "; FETCH_SIZE / WRITE_SIZE are KiB, and on gfx950 "
               "FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) per launch"}
def dbs(sub):
    return glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
for f in dbs("trace"):
    con = sqlite3.connect(f)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    kt = next((t for t in tabs if t == "kernels" or t.startswith("kernels")), None)
    rows = []
    try:
        q = f"select name, count(*), avg(duration), min(duration), max(duration) from {kt} group by name order by sum(duration) desc"
        rows = list(con.execute(q))
    except Exception as e:
        res["trace_error"] = str(e)[:200] + " tables: " + ",".join(tabs)[:400]
    res["kernel_stats"] = [{"kernel": r[0][:120], "calls": r[1], "avg_us": r[2] / 1e3, "min_us": r[3] / 1e3, "max_us": r[4] / 1e3} for r in rows[:8]]
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in dbs("pmc_" + ctr):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        ct = next((t for t in tabs if t.startswith("counters_collection")), None)
        if ct:
            for kname, cnt, avg in con.execute(f"select kernel_name, count(*), avg(value) from {ct} where kernel_name like '%k_sys64%' and counter_name = ? group by kernel_name", (ctr,)):
                res[ctr] = {"dispatches": cnt, "avg_per_launch": avg}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["hbm_bytes_per_launch"] = 1024.0 * (2 * res["FETCH_SIZE"]["avg_per_launch"] + res["WRITE_SIZE"]["avg_per_launch"])
    res["algorithmic_bytes_per_launch"] = 40.0 * 256 * 250 * 1025 * 2 if wide else 40.0 * 256 * 500 * 513 * 4
json.dump(res, open(os.path.join(out, f"{tag}_sys64_profile.json"), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $OUT/trace $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
