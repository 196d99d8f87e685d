# round 5: eval-mode flow (K9, non-blocking code upload, ICP queries in cell order + 64-workgroup reduction) against the plain PLY flow and the pipeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_gpu_icp.py tests/test_gpu_chamfer.py tests/test_experiment_io.py -x -q -m gpu 2>&1 | tail -3
(
echo "# tools/time_reconstruct_files.py 256 24 [eval], 3 runs each, two rounds on one box (MI355X, 1 GPU, synthetic nerf3 decoder, PLY export on);"
echo "# round 5 final: K9 (surface sampling + normalisation in five launches), non-blocking code upload (CodeUploader), ICP queries in cell order, 64-workgroup ICP reduction"
for round in 1 2; do
  echo "== round $round"
  ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 eval 2>/dev/null | grep -v "^$"
  ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 2>/dev/null | grep -v "^$"
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('sample pipeline without files (bench.py, 24 steps): %.2f ms/step' % b['ms_per_step'])"
done
python tools/time_icp.py 2>&1 | grep lanes | sed 's/lanes 8: /stand-alone ICP 30k x 30k (tools\/time_icp.py): /'
) | tee gpurun_out/r5/eval_flow_timing.txt
R=r5 SAMPLES=24 bash tools/trace_eval_flow.sh > /dev/null 2>&1; cat gpurun_out/r5/trace_eval/summary.txt
