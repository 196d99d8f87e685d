cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_fullsize.py -q -s > gpurun_out/r3/fullsize.log 2>&1; echo rc=$?
grep -E "surfaces with|other sign|passed|failed" gpurun_out/r3/fullsize.log | tail -40
