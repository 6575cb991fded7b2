#!/bin/bash
# per-kernel average times of one bench run (rocprofv3 kernel trace)
cd /root/repo; export TMPDIR=/tmp
mkdir -p gpurun_out/prof; rm -rf gpurun_out/prof/t2
rocprofv3 --kernel-trace -d gpurun_out/prof/t2 -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3 > /dev/null 2> gpurun_out/prof/t2.err
python3 - <<'EOF'
import sqlite3,glob
f=glob.glob("gpurun_out/prof/t2/**/*.db",recursive=True)
con=sqlite3.connect(f[0])
for r in con.execute("select name,count(*),avg(duration)/1e3 from kernels group by name order by 3 desc limit 10"): print("%-100s %3d %10.1f us"%(r[0][:100],r[1],r[2]))
EOF
