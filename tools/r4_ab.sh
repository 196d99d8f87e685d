#!/bin/bash
# same-box A/B of two builds of the library (tools/bin/libalignsdf_hip_<A>.so / _<B>.so): interleaved bench runs
A=$1; B=$2
for rep in 1 2 3; do
  for v in $A $B; do
    cp tools/bin/libalignsdf_hip_$v.so alignsdf_amd/csrc/libalignsdf_hip.so
    python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-other-math --no-other-configs --no-other-sweeps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'ms/step %.3f' % d['ms_per_step'], 'launch %.3f' % r['launch_ms'], 'clocks %.3fM' % (r['shader_clocks_per_launch']/1e6), 'GHz %.3f' % r['shader_clock_ghz'], 'refused', d['sweeps']['refused_sweeps'], d['parity_in_run'].get('against_reference_runs') and [x['V_F_equal_reference'] for x in d['parity_in_run']['against_reference_runs']])"
  done
done
