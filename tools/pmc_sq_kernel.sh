#!/bin/bash
# SQ / LDS counters of ONE kernel: three rocprofv3 --pmc passes of <= 8 counters each (counters only with --kernel-trace; no
# other trace domain) around a command that launches it.
# usage (GPU box, repo root):  bash tools/pmc_sq_kernel.sh <tag> <kernel-name-substring> <waves-per-workgroup> <command ...>
#   e.g. bash tools/pmc_sq_kernel.sh r03_online k_online 9 python tools/time_stage.py --stage online --reps 2
TAG=$1; KERN=$2; WPW=$3; shift 3
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONPATH=.
OUT=gpurun_out/pmcsq_$TAG
mkdir -p $OUT
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $OUT/pass$i -o $TAG -- "$@" > $OUT/pass$i.out 2> $OUT/pass$i.err
done
python3 - $OUT $TAG "$KERN" $WPW "$*" <<'PY'
import glob, json, os, sqlite3, sys
out, tag, kern, wpw, cmd = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
res = {"_how": "tools/pmc_sq_kernel.sh: rocprofv3 --kernel-trace --pmc <counters> -- %s, three passes; dispatches whose kernel name contains '%s'; "
               "sums over the chip, averaged over the dispatches seen.  SQ cycle counters tick once per 4 clocks." % (cmd, kern), "counters": {}}
names = set()
for d in sorted(glob.glob(os.path.join(out, "pass*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        ct = next((t for t in tabs if t == "counters_collection" or t.startswith("counters_collection")), None)
        if not ct:
            continue
        for kname, name, cnt, avg in con.execute(f"select kernel_name, counter_name, count(*), avg(value) from {ct} where kernel_name like ? group by kernel_name, counter_name", ("%" + kern + "%",)):
            names.add(kname[:160])
            res["counters"][name] = {"dispatches": cnt, "avg_per_dispatch": avg}
res["kernels"] = sorted(names)
c = {k: v["avg_per_dispatch"] for k, v in res["counters"].items()}
der = {}
if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
    wg = c["SQ_WAVES"] / wpw
    der["workgroups"] = wg
    wave_quads = c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]          # quad-cycles a wave is resident = kernel duration in quad-cycles (persistent waves)
    der["kernel_clocks_from_wave_cycles"] = 4 * wave_quads
    cu_quads = wave_quads * min(wg, 256.0)                        # CU-time available (one workgroup per CU)
    der["valu_busy_fraction_of_simd_time"] = c.get("SQ_ACTIVE_INST_VALU", 0) / (4 * cu_quads)
    der["lds_instruction_active_fraction_per_cu"] = c.get("SQ_ACTIVE_INST_LDS", 0) / cu_quads
    # SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array cycles (4 per ds_read_b128), not quad-cycles
    der["lds_array_active_fraction_per_cu"] = c.get("SQ_LDS_IDX_ACTIVE", 0) / (4 * cu_quads)
    der["lds_bank_conflict_fraction_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0))
    der["wave_fraction_waiting_any"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    der["wave_fraction_issue_stalled"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    der["wave_fraction_issuing"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    for k, nm in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_SALU", "salu"), ("SQ_INSTS_VMEM", "vmem")):
        if k in c:
            der[nm + "_instructions_per_workgroup"] = c[k] / wg
res["derived"] = der
json.dump(res, open(os.path.join(out, f"{tag}_pmc_sq.json"), "w"), indent=1)
print(json.dumps(res["derived"], indent=1))
PY
rm -rf $OUT/pass1 $OUT/pass2 $OUT/pass3
ls -la $OUT
