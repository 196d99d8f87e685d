cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_default_sweeps.py tests/test_gpu_coarse_box.py -x -q > gpurun_out/r3/t1.log 2>&1; echo "t1 rc=$?"
tail -n 30 gpurun_out/r3/t1.log
python -m pytest tests -m gpu -q > gpurun_out/r3/t_all.log 2>&1; echo "all rc=$?"
tail -n 40 gpurun_out/r3/t_all.log
python bench.py --steps 8 --warmup 2 > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/r3/bench1.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 3 gpurun_out/r3/smoke.log
