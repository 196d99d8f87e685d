#!/bin/bash
# tools/bin/libalignsdf_hip_<name>.so = the shipped library with ONE translation unit rebuilt under extra flags:
#   build_k1s_variant.sh <name> [-u unit.hip] [flags...]        (default unit: k1s_kernels.hip; the unit's shipped per-unit flags are kept)
name=$1; shift
unit=k1s_kernels.hip
if [ "$1" = "-u" ]; then unit=$2; shift 2; fi
cd "$(dirname "$0")/../alignsdf_amd/csrc"
tu=$(python3 -c "import sys; sys.path.insert(0, '../..'); from alignsdf_amd.build_native import TU_FLAGS; print(' '.join(TU_FLAGS.get('$unit', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $tu "$@" -c $unit -o /tmp/variant_$name.o || exit 1
objs=""
for u in decoder k1_kernels k1_cls_kernels k1h_kernels k1h_nerf_kernels k1s_kernels k1s_nerf_kernels mc33 icp mesh_cc surface_sample; do
  if [ "$u.hip" = "$unit" ]; then objs="$objs /tmp/variant_$name.o"; else objs="$objs $u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../tools/bin/libalignsdf_hip_$name.so
