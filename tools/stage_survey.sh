export PYTHONPATH=.
for cfg in "2048 256 64" "1024 64 64" "1000 200 64" "2048 512 64" "4096 1024 32" "1024 128 64"; do
  set -- $cfg
  for st in nofuture online batch; do
    python -W ignore tools/time_stage.py --stage $st --fsize $1 --fshift $2 --B $3 --T 500 --reps 2 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
