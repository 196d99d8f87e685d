# full GPU suite + smoke + the small-lattice figures.  TAG names the output files
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
T=${TAG:-x}
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5/gputests_$T.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r5/smoke_$T.txt
python tools/box_sweep_timing.py 64 hand > gpurun_out/r5/box_sweep_timing_$T.txt 2>&1
python tools/box_sweep_timing.py 128 both >> gpurun_out/r5/box_sweep_timing_$T.txt 2>&1
for cfg in "64 hand" "128 both"; do
  set -- $cfg
  for rep in 1 2 3; do
    python bench.py --grid $1 --branches $2 --steps 128 --warmup 8 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('N=$1 $2: %.4f ms/step' % b['ms_per_step'])" >> gpurun_out/r5/small_bench_$T.txt
  done
done
tail -3 gpurun_out/r5/gputests_$T.txt; cat gpurun_out/r5/smoke_$T.txt; grep "audit 65536" gpurun_out/r5/box_sweep_timing_$T.txt; cat gpurun_out/r5/small_bench_$T.txt
