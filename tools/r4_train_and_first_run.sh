#!/bin/bash
# round 4: train the grasp decoders on the GPU box (PyTorch-ROCm), then the GPU suite and a bench line - how tests/golden/grasp_decoder_*.npz were made
set -x
mkdir -p gpurun_out/r4
python tests/golden/train_grasp_decoders.py grasp3 grasp9 --device cuda --steps 20000 --per-scene 2048 --out gpurun_out/r4 > gpurun_out/r4/train.log 2>&1
tail -40 gpurun_out/r4/train.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r4/pytest1.log
cat gpurun_out/r4/pytest1.log
timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r4/bench1.json 2> gpurun_out/r4/bench1.err
tail -c 3000 gpurun_out/r4/bench1.json
