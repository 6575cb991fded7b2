#!/bin/bash
for v in "$@"; do
  echo "=== $v"
  LWS_HIP_LIB=/root/repo/lws_amd/variants/lib_$v.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(d['value']/1e9,'Gbin-it/s', d['ms_per_step'],'ms/step kernel', r['kernel_ms_per_step'],'ms frac',r['frac'], r['kernel'], 'resid', d['extra']['residual_db_after'])"
done
