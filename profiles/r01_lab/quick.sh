#!/bin/bash
# quick check of the default library: dense bench line (short) + systolic/parity tests
cd /root/repo
timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('ms/step',d['ms_per_step'],'kernel', r['kernel_ms_per_step'],'frac', r['frac'],'resid', d['extra']['residual_db_after'])"
[ "$1" = "notest" ] || timeout 600 python -m pytest tests/test_gpu_systolic.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
