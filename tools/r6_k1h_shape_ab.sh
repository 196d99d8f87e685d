#!/bin/bash
# Round 6: the split-half kernel on 16x16x32 MFMAs (the W form, default) against the 32x32x16 form (ASDF_K1H_SHAPE=32), through bench.py
# with the product's default (ordinary sweeps), interleaved on one box.
cd "$(dirname "$0")/.."
O=gpurun_out/r6; mkdir -p $O
for r in 1 2 ${ROUNDS:-3}; do
  for v in 32 16; do
    ASDF_K1H_SHAPE=$v python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-other-math --no-other-configs --sustained 0 --details /tmp/ab_details.json > /tmp/ab_line.json 2>/tmp/ab_err.txt || tail -5 /tmp/ab_err.txt
    python - <<PY
import json
d = json.loads([l for l in open('/tmp/ab_line.json') if l.startswith('{')][-1]); r = d['roofline']; p = d['config'].get('parity_in_run', {})
print('shape %s' % '$v', 'ms/step %.3f' % d['ms_per_step'], 'meshes/s %.3f' % d['value'], 'kernel', r['kernel'], 'launch %.3f ms' % r['launch_ms'], 'frac %.4f' % r['frac'],
      'GHz', r.get('shader_clock_ghz'), 'busy', r.get('pipe_busy'), 'parity', {k: p[k] for k in p if 'max_abs' in k or 'reference' in k or 'sign' in k})
PY
  done
done 2>&1 | tee $O/k1h_shape_${TAG:-a}.txt
