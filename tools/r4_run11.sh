#!/bin/bash
mkdir -p gpurun_out/r4
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4/pytest11.log
cat gpurun_out/r4/pytest11.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
