#!/bin/bash
# Round 6: the two forms of K1h on the SMALL lattices (configs[0] hand-only N = 64, configs[1] N = 128), interleaved on one box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6
for r in 1 2 3; do
  for v in 32 16; do
    for cfg in "--grid 64 --branches hand" "--grid 128"; do
      ASDF_K1H_SHAPE=$v python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps --sustained 0 --details /tmp/ab_details.json > /tmp/ab_line.json 2>/tmp/ab_err.txt || tail -3 /tmp/ab_err.txt
      python - <<PY
import json
d = json.loads([l for l in open('/tmp/ab_line.json') if l.startswith('{')][-1]); r = d['roofline']
print('shape %s' % '$v', '%-26s' % '$cfg', 'ms/step %.4f' % d['ms_per_step'], 'kernel', r['kernel'], 'launch %.4f ms' % r['launch_ms'], 'GHz', r.get('shader_clock_ghz'))
PY
    done
  done
done 2>&1 | tee gpurun_out/r6/k1h_shape_small.txt
