#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_default_sweeps.py::test_grasp_family_at_other_sizes_and_single_branches -q -m gpu -s 2>&1 | tail -30
