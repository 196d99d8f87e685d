# Round 5: what the driver will run, on one box: the GPU tests, smoke, the default bench line (+ its side file); then the kernel statistics
# of the same command under rocprofv3 (default sweeps and every-voxel sweeps) and the small-lattice traces.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r5/gputests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r5/bench_default_N256_details.json > gpurun_out/r5/bench_default_N256.json 2> gpurun_out/r5/bench_default_N256.err
wc -c gpurun_out/r5/bench_default_N256.json
O=$GRAFT_REPO_ROOT/gpurun_out/r5/stats; rm -rf $O; mkdir -p $O
ARGS="--no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 8 --warmup 2 $ARGS --details $O/bench_under_kernel_trace_details.json > $O/bench_under_kernel_trace.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_exact -- python bench.py --steps 4 --warmup 2 --coarse exact --fine exact $ARGS --details $O/bench_exact_under_kernel_trace_details.json > $O/bench_exact_under_kernel_trace.json 2> $O/stats_exact.err
for tag in stats stats_exact; do cp $(find $O/$tag -name "*kernel_stats.csv" | head -1) $O/kernel_$tag.csv; done
head -8 $O/kernel_stats.csv | cut -c1-150
( echo "# tools/trace_small_lattice.sh on one MI355X (rocprofv3 --kernel-trace over bench.py), round 5 after the cluster form / audit stream / fused launches"
  echo "== hand-only, N = 64 (configs[0])"; R=r5 bash tools/trace_small_lattice.sh 64 hand 64 2>/dev/null
  echo; echo "== hand + object, N = 128 (configs[1])"; R=r5 bash tools/trace_small_lattice.sh 128 both 32 2>/dev/null ) > gpurun_out/r5/small_lattice_traces.txt
python -c "
import json; d=json.load(open('gpurun_out/r5/bench_default_N256.json')); print(json.dumps(d, indent=1)[:6500])"
