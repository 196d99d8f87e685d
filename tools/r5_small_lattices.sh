# Round 5, the small lattices (configs[0] hand-only N = 64, configs[1] N = 128) on one box:
#   profiles/r05_box_sweep_timing.txt   back-to-back box / band sweeps, untraced: audit beside or in line, cluster form or short form
#   profiles/r05_small_lattice_bench.txt   bench.py over 64 and 256 (192) steps, and the bare pipeline's per-sample times
# gpurun -- 'bash tools/r5_small_lattices.sh'   ->  gpurun_out/r5/{box_sweep_timing,small_lattice_bench}.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
( python tools/box_sweep_timing.py 64 hand; python tools/box_sweep_timing.py 128 both ) 2>&1 | grep "N=" > gpurun_out/r5/box_sweep_timing.txt
O=gpurun_out/r5/small_lattice_bench.txt; : > $O
for cfg in "64 hand 64" "64 hand 256" "128 both 64" "128 both 192"; do
  set -- $cfg
  for rep in 1 2; do
    python bench.py --grid $1 --branches $2 --steps $3 --warmup 8 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('N=$1 $2 steps $3: %.4f ms/step' % b['ms_per_step'])" >> $O
  done
done
python tools/per_sample_times.py 64 hand 300 2>&1 | grep "N=" >> $O
python tools/per_sample_times.py 128 both 200 2>&1 | grep "N=" >> $O
cat gpurun_out/r5/box_sweep_timing.txt $O
