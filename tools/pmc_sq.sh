#!/bin/bash
# SQ / LDS counters of the headline kernel (BASELINE config 2 alone), three rocprofv3 --pmc passes of <= 8 counters each
# (counters only with --kernel-trace; no other trace domain).  usage (on the GPU box, repo root):  bash tools/pmc_sq.sh r02
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/pmcsq_$TAG
mkdir -p $OUT
P1="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES"
P2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $OUT/pass$i -o $TAG -- python bench.py --no-extras --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-pass > /dev/null 2> $OUT/pass$i.err
done
python3 - $OUT $TAG <<'PY'
import glob, json, os, sqlite3, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"_how": "tools/pmc_sq.sh: rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --no-extras --steps 2 --warmup 1 --no-cpu-baseline --no-traffic-pass, "
               "three passes; k_systolic<4,5,hann> dispatches only (256 workgroups x 8 waves, 100 dense sweeps); sums over the chip, "
               "averaged over the dispatches seen.  SQ cycle counters tick once per 4 clocks.", "counters": {}}
for d in sorted(glob.glob(os.path.join(out, "pass*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
        ct = next((t for t in tabs if t == "counters_collection" or t.startswith("counters_collection")), None)
        if not ct:
            continue
        for name, cnt, avg in con.execute(f"select counter_name, count(*), avg(value) from {ct} where kernel_name like '%k_systolic%' group by counter_name"):
            res["counters"][name] = {"dispatches": cnt, "avg_per_dispatch": avg}
c = {k: v["avg_per_dispatch"] for k, v in res["counters"].items()}
der = {}
if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
    # SQ_BUSY_CYCLES sums over the SEs/XCDs; per-SIMD time = wave cycles / waves-per-SIMD is the sturdier base
    pass
if "SQ_WAVE_CYCLES" in c and "SQ_WAVES" in c:
    simd_cycles = c["SQ_WAVE_CYCLES"] / 2.0          # two waves per SIMD, resident for the whole kernel
    der["valu_busy_fraction_of_simd_time"] = c.get("SQ_ACTIVE_INST_VALU", 0) / simd_cycles
    cu_cycles = c["SQ_WAVE_CYCLES"] / 8.0             # eight waves per CU
    der["lds_instruction_active_fraction_per_cu"] = c.get("SQ_ACTIVE_INST_LDS", 0) / cu_cycles
    der["lds_index_active_fraction_per_cu"] = c.get("SQ_LDS_IDX_ACTIVE", 0) / cu_cycles
    der["lds_bank_conflict_fraction_of_cycles_per_cu"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / cu_cycles
pairs = 256 * 500 * 513 * 100 / 2 / 64.0             # pairs of bins per lane-wave ... per workgroup-wave: bins / (2 bins x 64 lanes)
for k, nm in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_LDS", "lds"), ("SQ_INSTS_SALU", "salu")):
    if k in c:
        der[nm + "_instructions_per_pair_of_bins_and_wave"] = c[k] / pairs / (8.0 / 7.0) if False else c[k] / (256 * 500 * 513 * 100 / 128.0)
res["derived"] = der
json.dump(res, open(os.path.join(out, f"{tag}_pmc_sq_counters.json"), "w"), indent=1)
print(json.dumps(res["derived"], indent=1))
PY
rm -rf $OUT/pass1 $OUT/pass2 $OUT/pass3
ls -la $OUT
