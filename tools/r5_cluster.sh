# round 5: the cluster form of the short-list kernel and the audit beside the candidates - parity, then the small-lattice traces
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_short_list.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5/cluster_tests.txt
for cfg in "64 hand" "128 both"; do
  set -- $cfg
  python bench.py --grid $1 --branches $2 --steps 64 --warmup 8 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('N=$1 $2: %.4f ms/step' % b['ms_per_step'])" | tee -a gpurun_out/r5/cluster_bench.txt
  ASDF_AUDIT_INLINE=1 python bench.py --grid $1 --branches $2 --steps 64 --warmup 8 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('N=$1 $2 (audit in line): %.4f ms/step' % b['ms_per_step'])" | tee -a gpurun_out/r5/cluster_bench.txt
done
R=r5 bash tools/trace_small_lattice.sh 64 hand 64 > gpurun_out/r5/cluster_trace_64.txt 2>&1
R=r5 bash tools/trace_small_lattice.sh 128 both 32 > gpurun_out/r5/cluster_trace_128.txt 2>&1
tail -30 gpurun_out/r5/cluster_trace_64.txt
