#!/bin/bash
# Round 5: the split-half kernel K1h, same box, interleaved.
#   (1) tools/k1h_ablate.hip variants (tools/bin/k1h_<v>, built with -DASDF16_SEGMENT_TIMES) on the product's weights: ms, clock, cycles per layer
#   (2) library builds (tools/bin/libalignsdf_hip_<v>.so) through bench.py with ORDINARY sweeps (every voxel on K1h)
#   VARIANTS="r5base r5mix" LIBS="ship r5a" gpurun -- 'bash tools/r5_k1h_ab.sh'
cd "$(dirname "$0")/.."
O=gpurun_out/r5; mkdir -p $O
{
for r in 1 2 ${ROUNDS:-3}; do
  for v in ${VARIANTS:-r5base r5mix r5mixp2}; do
    echo "== $v round $r"; tools/bin/k1h_$v 256 tools/bin/k1h_nerf3.bin
  done
done
} 2>&1 | tee $O/k1h_ablate_${TAG:-a}.txt | grep -E "^==|ABL|one tile"
cp alignsdf_amd/csrc/libalignsdf_hip.so /tmp/lib_keep.so
{
for r in 1 2; do
  for v in ${LIBS:-ship r5a}; do
    cp tools/bin/libalignsdf_hip_$v.so alignsdf_amd/csrc/libalignsdf_hip.so
    python bench.py --coarse exact --fine exact --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-other-math --no-other-configs --sustained 0 --details /tmp/k1h_ab_details.json > /dev/null 2>&1
    python -c "
import json
d=json.load(open('/tmp/k1h_ab_details.json')); r=d['roofline']; p=d['parity_in_run']
print('%-6s' % '$v', 'ms/step %.3f' % d['ms_per_step'], 'kernel', r['kernel'], 'launch %.3f' % r['launch_ms'], 'frac %.4f' % r['frac'], 'GHz', r.get('shader_clock_ghz'), 'busy', r.get('pipe_busy'),
      'vol', p.get('volumes_f16x3_vs_f32'), 'ref', [x['V_F_equal_reference'] for x in p.get('against_reference_runs', [])])"
  done
done
} 2>&1 | tee $O/k1h_bench_${TAG:-a}.txt
cp /tmp/lib_keep.so alignsdf_amd/csrc/libalignsdf_hip.so
