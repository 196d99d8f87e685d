// Lewiner marching cubes on gfx950 (placeholder until the kernels land).
#include <hip/hip_runtime.h>
#include "../../include/alignsdf_hip.h"
extern "C" {
int asdf_mc_workspace_bytes(int32_t, int32_t, int32_t, size_t*) { return ASDF_EINVAL; }
int asdf_mc_count(const float*, int32_t, int32_t, int32_t, float, void*, size_t, uint32_t*, uint32_t*, void*) { return ASDF_EINVAL; }
int asdf_mc_emit(const float*, int32_t, int32_t, int32_t, float, void*, size_t, float*, int32_t*, void*) { return ASDF_EINVAL; }
}
