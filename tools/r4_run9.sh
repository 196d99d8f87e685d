#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s 2>&1 | tail -80 > gpurun_out/r4/pytest9.log
cat gpurun_out/r4/pytest9.log
timeout 900 python -m pytest "tests/test_gpu_default_sweeps.py::test_default_meshes_are_the_ordinary_sweeps_meshes_all_64_samples_n256" -q -m gpu -s 2>&1 | tail -12 > gpurun_out/r4/pytest9b.log
cat gpurun_out/r4/pytest9b.log
