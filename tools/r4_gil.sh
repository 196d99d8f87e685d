#!/bin/bash
for iv in 5e-4 1e-4 2e-5 5e-4 1e-4 2e-5; do
  echo "== switch interval $iv"
  ASDF_GIL_INTERVAL=$iv ASDF_TIMING_REPS=3 python tools/time_reconstruct_files.py 256 8 eval 2>/dev/null | grep "reconstruct(eval_mode)"
done
