# Round 5: what the driver will run, on one box: the GPU tests, smoke, the default bench line (+ its side file).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r5/gputests_final.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/r5/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --details gpurun_out/r5/bench_default_N256_details.json > gpurun_out/r5/bench_default_N256.json 2> gpurun_out/r5/bench_default_N256.err
wc -c gpurun_out/r5/bench_default_N256.json; python -c "
import json; d=json.load(open('gpurun_out/r5/bench_default_N256.json')); print(json.dumps(d, indent=1)[:7000])"
