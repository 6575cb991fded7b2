#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace + stats of the default bench command, and FETCH_SIZE / WRITE_SIZE PMC
# passes (separate runs, as MI355X_MICROARCH.md prescribes) of every BASELINE configuration.
# usage (on the GPU box, from the repo root):  bash tools/profile.sh r06
TAG=${1:-r06}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o $TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic-pass --extra-file $OUT/extra_traced.json > $OUT/bench_traced.json 2> $OUT/trace.err
# the headline workload alone (config 2 only): the per-kernel averages of this one are those of bench.py's roofline block
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_headline -o $TAG -- python bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-traffic-pass --extra-file $OUT/extra_scratch.json > $OUT/bench_traced_headline.json 2> $OUT/trace_headline.err
for cfg in 2 2-T1024 2-q2 2-q8 2-q3 2-frac 2-speech 2-q5 2-q8w 2-q16 2-l8 2-f501 2-f257 4shard 5 5-f16; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_${ctr}_$cfg -o $TAG -- python bench.py --config $cfg --no-extras --steps 1 --warmup 1 --no-cpu-baseline --no-traffic-pass --extra-file $OUT/extra_scratch.json > $OUT/pmc_line_$cfg.json 2> $OUT/pmc_${ctr}_$cfg.err
  done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_${ctr}_3 -o $TAG -- python bench.py --extras 3 --steps 1 --warmup 1 --no-cpu-baseline --no-traffic-pass --extra-file $OUT/extra_3.json > /dev/null 2> $OUT/pmc_${ctr}_3.err
done
python3 tools/collect_profiles.py $OUT $OUT/summary $TAG
# the rocpd databases are tens of MiB each: only the summaries (and the error logs) travel back
rm -rf $OUT/trace $OUT/trace_headline $OUT/pmc_FETCH_SIZE_* $OUT/pmc_WRITE_SIZE_*
find $OUT -name '*.err' -size +64k -delete
ls -la $OUT/summary
