#!/bin/bash
# same-box A/B of two builds of the library (tools/bin/libalignsdf_hip_<A>.so / _<B>.so): interleaved bench runs incl. the ordinary sweeps and the fp32 chain
A=$1; B=$2
for rep in 1 2; do
  for v in $A $B; do
    cp tools/bin/libalignsdf_hip_$v.so alignsdf_amd/csrc/libalignsdf_hip.so
    python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', 'ms/step %.3f' % d['ms_per_step'], 'K1s launch %.3f' % r['launch_ms'], 'GHz %.3f' % r['shader_clock_ghz'], '| ordinary sweeps %.2f ms/step, K1h launch %s' % (d['other_sweeps']['ms_per_step'], d['other_sweeps'].get('launch_ms')), '| f32 %.2f ms/step launch %s' % (d['other_math']['ms_per_step'], d['other_math'].get('launch_ms')), 'refused', d['sweeps']['refused_sweeps'], d['parity_in_run']['against_fp32_chain'], d['parity_in_run']['against_ordinary_sweeps_f16x3'])"
  done
done
