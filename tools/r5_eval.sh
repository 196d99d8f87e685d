# round 5: eval-mode flow after K9 (device surface sampling + normalisation in four launches) and the non-blocking code upload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_icp.py tests/test_experiment_io.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5/eval_tests.txt
(
for round in 1 2; do
  echo "== round $round"
  ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 eval 2>/dev/null | grep -v "^$"
  ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 2>/dev/null | grep -v "^$"
done
python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('sample pipeline without files (bench.py, 24 steps): %.2f ms/step' % b['ms_per_step'])"
) | tee gpurun_out/r5/eval_flow_timing.txt
R=r5 SAMPLES=24 bash tools/trace_eval_flow.sh > /dev/null 2>&1; cat gpurun_out/r5/trace_eval/summary.txt
