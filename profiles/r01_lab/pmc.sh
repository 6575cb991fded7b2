#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|GRBM_[A-Z_]+" | sort -u | tr '\n' ' ' > gpurun_out/pmc/counters.txt
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc/p1 -o r -- $B > /dev/null 2> gpurun_out/pmc/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d gpurun_out/pmc/p2 -o r -- $B > /dev/null 2> gpurun_out/pmc/p2.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_WAVE32_LDS -d gpurun_out/pmc/p3 -o r -- $B > /dev/null 2> gpurun_out/pmc/p3.err
python3 - <<'PY'
import sqlite3,glob
for f in sorted(glob.glob('gpurun_out/pmc/p*/r_results.db')):
    con=sqlite3.connect(f); cur=con.cursor()
    for r in cur.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k_systolic%' group by counter_name"): print(f.split('/')[2], r)
PY
tail -3 gpurun_out/pmc/p1.err gpurun_out/pmc/p3.err
