cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r5
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5/gputests_i.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/r5/smoke_i.txt
(
echo "# tools/time_reconstruct_files.py 256 24 [eval], 3 runs each, two rounds on one box (MI355X, 1 GPU, synthetic nerf3 decoder, PLY export on);"
echo "# round 5 after K9 (surface sampling + normalisation in four launches) and the non-blocking code upload (CodeUploader)"
for round in 1 2; do
  echo "== round $round"
  ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 eval 2>/dev/null | grep -v "^$"
  ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 24 2>/dev/null | grep -v "^$"
  python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-other-math --no-other-sweeps --no-other-configs --sustained 0 2>/dev/null | tail -1 | python -c "import sys, json; b = json.loads(sys.stdin.read()); print('sample pipeline without files (bench.py, 24 steps): %.2f ms/step' % b['ms_per_step'])"
done
echo "== ASDF_COARSE=exact ASDF_FINE=exact (ordinary sweeps in both passes)"
ASDF_COARSE=exact ASDF_FINE=exact ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 8 eval 2>/dev/null | grep -v "^$"
ASDF_COARSE=exact ASDF_FINE=exact ASDF_TIMING_REPS=3 ASDF_TIMING_FLOW_ONLY=1 python tools/time_reconstruct_files.py 256 8 2>/dev/null | grep -v "^$"
) > gpurun_out/r5/eval_flow_timing.txt
R=r5 SAMPLES=24 bash tools/trace_eval_flow.sh > /dev/null 2>&1
tail -3 gpurun_out/r5/gputests_i.txt; cat gpurun_out/r5/smoke_i.txt gpurun_out/r5/eval_flow_timing.txt; head -3 gpurun_out/r5/trace_eval/summary.txt; tail -3 gpurun_out/r5/trace_eval/summary.txt
