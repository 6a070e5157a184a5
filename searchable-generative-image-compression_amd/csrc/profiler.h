// Per-launch timing handle (include/sgic.h: sgic_profiler_*), shared by the files that launch GEMM-class kernels.
#pragma once
#include <utility>
#include <vector>

#include <hip/hip_ext.h>

#include "common.h"

struct sgic_profiler {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  int n = -1;  // -1: closed
};


// next event pair of the open window (the pool grows on demand), or null when the launch carries no open profiler
static inline const std::pair<hipEvent_t, hipEvent_t> *prof_next(const sgic_launch_opts *o) {
  sgic_profiler *p = o ? o->profiler : nullptr;
  if (!p || p->n < 0) return nullptr;
  while (p->n >= (int)p->pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return nullptr;
    if (hipEventCreate(&b) != hipSuccess) return nullptr;
    p->pool.emplace_back(a, b);
  }
  return &p->pool[p->n++];
}

