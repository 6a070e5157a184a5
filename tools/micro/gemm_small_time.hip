// Diagnostic: GPU time of small GEMMs by C++ launches (no Python wrapper in the loop), 64x64 kernel (mode 13) vs latency kernel (15).
// Derived from gemm_stamps.hip:  Builds csrc/gemm.hip with per-workgroup stamps and prints, per
// shape and tile mode: launch duration by events, first-start -> last-end span, and per workgroup the time in main loops
// vs epilogues (wave 0) and the spread of workgroup end times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/gemm_stamps tools/micro/gemm_stamps.hip && /tmp/gemm_stamps
#define GEMM_STAMPS
#include <algorithm>
#include <vector>
#include <stdarg.h>
#include "../../searchable-generative-image-compression_amd/csrc/common.h"
namespace sgic {
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
}  // namespace sgic
#include "../../searchable-generative-image-compression_amd/csrc/gemm.hip"

int main() {
  struct Shape { int M, N, K, res, act; } shapes[] = {{289, 1024, 1024, 1, 0}, {289, 1024, 4096, 1, 0}, {289, 3072, 1024, 0, 0}, {289, 4096, 1024, 0, 1}, {545, 768, 3072, 1, 0}, {545, 2304, 768, 0, 0},
                                                      {256, 768, 768, 1, 0}, {1600, 768, 768, 1, 0}, {2048, 128, 256, 1, 0}, {64, 64, 512, 0, 0}};
  const int modes[] = {13, 15};
  for (auto sh : shapes) {
    float *A, *W, *C, *R, *bias;
    hipMalloc(&A, (size_t)sh.M * sh.K * 4);
    hipMalloc(&W, (size_t)sh.N * sh.K * 4);
    hipMalloc(&C, (size_t)sh.M * sh.N * 4);
    hipMalloc(&R, (size_t)sh.M * sh.N * 4);
    hipMalloc(&bias, (size_t)sh.N * 4);
    std::vector<float> h((size_t)std::max(sh.M, sh.N) * sh.K);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), (size_t)sh.M * sh.K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 4, hipMemcpyHostToDevice);
    hipMemset(R, 0, (size_t)sh.M * sh.N * 4);
    hipMemset(bias, 0, (size_t)sh.N * 4);
    for (int mode : modes) {
      sgic_launch_opts o{mode, 0, nullptr};
      auto run = [&]() {
        return sgic_gemm_f32(A, sh.K, W, sh.K, bias, sh.res ? R : nullptr, sh.N, C, sh.N, sh.M, sh.N, sh.K, sh.act, 0, 0, 0, 0, &o, nullptr);
      };
      for (int rep = 0; rep < 3; rep++) run();
      hipDeviceSynchronize();
      long long *dst;
      hipGetSymbolAddress((void **)&dst, HIP_SYMBOL(gemm_stamps));
      hipMemset(dst, 0, sizeof(long long) * 8 * 4096);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int rep = 0; rep < 20; rep++) run();
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20;
      printf("(%d,%d,%d,res=%d,act=%d) mode %d: %.1f us per launch, %.1f TF\n", sh.M, sh.N, sh.K, sh.res, sh.act, mode, ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms / 1e9);
      continue;
      std::vector<long long> st(8 * 4096);
      hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
      long long t0 = -1, t1 = 0;
      std::vector<double> ends, starts, loop, epi, life;
      double tiles = 0, loop_sum = 0, epi_sum = 0;
      for (int b = 0; b < 4096; b++) {
        const long long *s = &st[b * 8];
        if (!s[0] || !s[1]) continue;
        if (t0 < 0 || s[0] < t0) t0 = s[0];
        t1 = std::max(t1, s[1]);
      }
      for (int b = 0; b < 4096; b++) {
        const long long *s = &st[b * 8];
        if (!s[0] || !s[1]) continue;
        starts.push_back((s[0] - t0) / 100.0);
        ends.push_back((s[1] - t0) / 100.0);
        loop.push_back((double)s[2]);
        epi.push_back((double)s[3]);
        life.push_back((double)s[5]);
        tiles += s[4];
        loop_sum += s[2];
        epi_sum += s[3];
      }
      const size_t n = ends.size();
      if (!n) continue;
      std::sort(starts.begin(), starts.end());
      std::sort(ends.begin(), ends.end());
      std::sort(life.begin(), life.end());
      const double flops = 2.0 * sh.M * sh.N * sh.K;
      printf("(%d,%d,%d,res=%d,act=%d) mode %d: %.1f us (%.1f TF); stamped WGs %zu (first 4096 blocks); span %.1f us; start p50 %.1f p90 %.1f max %.1f | "
             "end p10 %.1f p50 %.1f p90 %.1f max %.1f us | per WG: tiles %.2f, cycles/tile loop %.0f epilogue %.0f (%.1f %% of loop+epi), life p50 %.0f cycles\n",
             sh.M, sh.N, sh.K, sh.res, sh.act, mode, ms * 1e3, flops / ms / 1e9, n, (t1 - t0) / 100.0, starts[n / 2], starts[n * 9 / 10], starts[n - 1],
             ends[n / 10], ends[n / 2], ends[n * 9 / 10], ends[n - 1], tiles / n, loop_sum / tiles, epi_sum / tiles, 100.0 * epi_sum / (loop_sum + epi_sum),
             life[n / 2]);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(R); hipFree(bias);
  }
  return 0;
}
