#!/bin/bash
# same-box comparison of several builds of the library (tools/bin/libalignsdf_hip_<name>.so), interleaved, incl. the ordinary sweeps (K1h)
for rep in 1 2; do
  for v in "$@"; do
    cp tools/bin/libalignsdf_hip_$v.so alignsdf_amd/csrc/libalignsdf_hip.so
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-other-configs --no-other-math 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-6s' % '$v', 'ms/step %.3f' % d['ms_per_step'], 'K1s launch %.3f' % r['launch_ms'], 'GHz %.3f' % r['shader_clock_ghz'], '| ordinary sweeps %.2f ms/step, K1h launch %.3f' % (d['other_sweeps']['ms_per_step'], d['other_sweeps']['launch_ms']), 'refused', d['sweeps']['refused_sweeps'], 'identical', d['parity_in_run']['against_ordinary_sweeps_f16x3']['vertices_identical'])"
  done
done
