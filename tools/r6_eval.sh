# round 6: the file-producing flows under the DEFAULT (ordinary sweeps) and under --fast (ASDF_FAST=1), with host-side codes read from
# pinned memory by the fold's staging launch (no side-stream copy, no runtime blit kernel) - timings + kernel traces
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r6
(
echo "# tools/time_reconstruct_files.py 256 24 [eval], 3 runs each (MI355X, 1 GPU, synthetic nerf3 decoder, PLY export on); round 6:"
echo "# host-side codes stay on the host (asdf_decoder_set_sample_host); default = ordinary sweeps, ASDF_FAST=1 = audited one-plane sweeps"
for fast in 0 1; do
  echo "== ASDF_FAST=$fast"
  ASDF_FAST=$fast ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 eval 2>/dev/null | grep -v "^$"
  ASDF_FAST=$fast ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 2>/dev/null | grep -v "^$"
  python bench.py --steps 24 --warmup 4 $( [ $fast = 1 ] && echo --fast ) --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('sample pipeline without files (bench.py, 24 steps): %.2f ms/step' % b['ms_per_step'])"
done
) | tee gpurun_out/r6/eval_flow_timing.txt
for fast in 0 1; do
  ASDF_FAST=$fast R=r6/fast$fast SAMPLES=24 bash tools/trace_eval_flow.sh > /dev/null 2>&1
  echo "== ASDF_FAST=$fast"; cat gpurun_out/r6/fast$fast/trace_eval/summary.txt
done | tee gpurun_out/r6/eval_flow_trace.txt
