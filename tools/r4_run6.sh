#!/bin/bash
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_coarse_box.py tests/test_gpu_refine.py tests/test_gpu_default_sweeps.py::test_audit_record_of_the_c_abi tests/test_bench_launch.py tests/test_gpu_pipeline.py tests/test_gpu_decoder.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r4/pytest6.log
cat gpurun_out/r4/pytest6.log
R=r4 bash tools/trace_small_lattice.sh 64 hand 64 > /dev/null 2>&1; cat gpurun_out/r4/trace_small_nerf3_64/summary.txt
R=r4 bash tools/trace_small_lattice.sh 128 both 32 > /dev/null 2>&1; cat gpurun_out/r4/trace_small_nerf3_128/summary.txt
rm -rf gpurun_out/r4/trace_small_*/t
