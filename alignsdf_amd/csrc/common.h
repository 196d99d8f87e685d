// Shared host-side helpers of libalignsdf_hip.so.
#pragma once
#include <hip/hip_runtime.h>

namespace asdf {
extern thread_local int g_last_hip_error;
}

// Evaluate a HIP runtime call; on failure record it and return ASDF_EHIP from the enclosing function.
#define ASDF_HIP(expr)                                         \
  do {                                                         \
    hipError_t asdf_e_ = (expr);                               \
    if (asdf_e_ != hipSuccess) {                               \
      asdf::g_last_hip_error = (int)asdf_e_;                   \
      return asdf_e_ == hipErrorOutOfMemory ? ASDF_ENOMEM : ASDF_EHIP; \
    }                                                          \
  } while (0)
