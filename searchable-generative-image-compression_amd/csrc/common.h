// Shared helpers for libsgic (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sgic.h"

namespace sgic {
void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SGIC_EHIP;
  }
  return SGIC_OK;
}
}  // namespace sgic

#define SGIC_HIP(call)                                                       \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) {                                                  \
      sgic::set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      return SGIC_EHIP;                                                      \
    }                                                                        \
  } while (0)

#define SGIC_REQUIRE(cond, msg)                               \
  do {                                                        \
    if (!(cond)) {                                            \
      sgic::set_error("%s: requirement failed: %s (%s)", __func__, #cond, msg); \
      return SGIC_EINVAL;                                     \
    }                                                         \
  } while (0)

static inline hipStream_t to_stream(sgic_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
