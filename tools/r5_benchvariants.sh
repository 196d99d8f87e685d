cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
O=gpurun_out/r5/bench_variants.txt; : > $O
for cfg in "64 hand 64" "64 hand 256" "128 both 64" "128 both 192"; do
  set -- $cfg
  for rep in 1 2; do
    python bench.py --grid $1 --branches $2 --steps $3 --warmup 8 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('N=$1 $2 steps $3: %.4f ms/step' % b['ms_per_step'])" >> $O
  done
done
python tools/per_sample_times.py 64 hand 300 2>&1 | grep "N=" >> $O
python tools/per_sample_times.py 128 both 200 2>&1 | grep "N=" >> $O
cat $O
