#!/bin/bash
# same-box comparison of several builds of the library (tools/bin/libalignsdf_hip_<name>.so), two interleaved rounds
for rep in 1 2; do
  for v in "$@"; do
    cp tools/bin/libalignsdf_hip_$v.so alignsdf_amd/csrc/libalignsdf_hip.so
    python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-6s' % '$v', 'ms/step %.3f' % d['ms_per_step'], 'launch %.3f' % r['launch_ms'], 'clocks %.3fM' % (r['shader_clocks_per_launch']/1e6), 'GHz %.3f' % r['shader_clock_ghz'], 'refused', d['sweeps']['refused_sweeps'], 'ref', [x['V_F_equal_reference'] for x in d['parity_in_run'].get('against_reference_runs', [])])"
  done
done
