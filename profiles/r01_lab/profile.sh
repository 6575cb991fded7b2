#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench command, and PMC passes for HBM traffic
set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.json
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/trace -o r01 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3 > gpurun_out/prof/bench_traced.json 2> gpurun_out/prof/trace.err
ls -R gpurun_out/prof/trace | head -20
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof/pmc_fetch -o r01 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3 > /dev/null 2> gpurun_out/prof/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof/pmc_write -o r01 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-default-schedule --no-config3 > /dev/null 2> gpurun_out/prof/pmc_write.err
ls -R gpurun_out/prof | head -40
python3 scratch/collect_profiles.py gpurun_out/prof gpurun_out/profiles_new r01
cp gpurun_out/bench_default.json gpurun_out/profiles_new/r01_bench_default.json
