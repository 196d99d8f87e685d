#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s 2>&1 | grep -v "^$" > gpurun_out/r4/fullsize_full.log
tail -3 gpurun_out/r4/fullsize_full.log
R=r4 bash tools/trace_small_lattice.sh 64 hand 64 > /dev/null 2>&1; cat gpurun_out/r4/trace_small_nerf3_64/summary.txt
R=r4 bash tools/trace_small_lattice.sh 128 both 32 > /dev/null 2>&1; cat gpurun_out/r4/trace_small_nerf3_128/summary.txt
rm -rf gpurun_out/r4/trace_small_*/t
bash tools/profile_bench_r4.sh > gpurun_out/r4/profile.log 2>&1
rm -rf gpurun_out/r4/prof/stats gpurun_out/r4/prof/pmc_[0-9]
head -12 gpurun_out/r4/prof/kernel_stats.csv | cut -c1-120
grep -A12 "sdf_mlp_f16p1_kernel" gpurun_out/r4/prof/pmc_summary.txt | head -40
