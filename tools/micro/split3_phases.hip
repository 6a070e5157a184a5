// Diagnostic: where does a split-GEMM launch spend its time?  Builds csrc/gemm_split.hip with S3_STAMPS and prints, per model
// shape and launch mode: launch duration (events, back-to-back), TFLOP/s, the shader clock the chip held (cycles of workgroup
// lifetimes / their s_memrealtime spans), the share of a workgroup's life in {prologue, main loop, epilogue}, the main loop
// against its MFMA floor (K/32 slices x MFMAs per wave per slice x 16 cycles x waves per SIMD), and the spread of workgroup end
// times (tail).  Random normal operands (the clock depends on the data).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/micro/bin/split3_phases tools/micro/split3_phases.hip
#define S3_STAMPS
#include <algorithm>
#include <random>
#include <stdarg.h>
#include <vector>
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
#include "../../searchable-generative-image-compression_amd/csrc/gemm_split.hip"

int main(int argc, char **argv) {
  struct Shape { int M, N, K, res, act; } shapes[] = {{9248, 4096, 1024, 0, 1}, {9248, 1024, 4096, 1, 0}, {9248, 3072, 1024, 0, 0}, {17440, 768, 3072, 1, 0},
                                                      {17440, 3072, 768, 0, 1}, {9248, 1024, 1024, 1, 0}, {8192, 768, 3072, 1, 0}, {8192, 2304, 768, 0, 0},
                                                      {16384, 4096, 1024, 0, 0}};
#ifdef S3_MICRO_BIG_ONLY
  const int modes[] = {1, 10};
#else
  const int modes[] = {1, 10, 2, 11, 12};
#endif
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto sh : shapes) {
    float *A, *W, *C, *R, *bias;
    uint16_t *Ap, *Wp;
    hipMalloc(&A, (size_t)sh.M * sh.K * 4);
    hipMalloc(&W, (size_t)sh.N * sh.K * 4);
    hipMalloc(&C, (size_t)sh.M * sh.N * 4);
    hipMalloc(&R, (size_t)sh.M * sh.N * 4);
    hipMalloc(&bias, (size_t)sh.N * 4);
    hipMalloc(&Ap, (size_t)sh.M * sh.K * 6);
    hipMalloc(&Wp, (size_t)sh.N * sh.K * 6);
    std::vector<float> h((size_t)std::max(sh.M, sh.N) * sh.K);
    for (auto &v : h) v = nd(rng);
    hipMemcpy(A, h.data(), (size_t)sh.M * sh.K * 4, hipMemcpyHostToDevice);
    for (auto &v : h) v = nd(rng) * 0.05f;
    hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 4, hipMemcpyHostToDevice);
    hipMemset(R, 0, (size_t)sh.M * sh.N * 4);
    hipMemset(bias, 0, (size_t)sh.N * 4);
    sgic_split3_pack_f32(A, sh.K, sh.M, sh.K, Ap, nullptr);   // the slice-major layout the model uses (round 3); -DROWMAJOR for the old one
    sgic_split3_pack_f32(W, sh.K, sh.N, sh.K, Wp, nullptr);
    for (int mode : modes) {
      if ((mode == 1 || mode == 10) && sh.N < 1024) continue;
      sgic_launch_opts o{mode, 0, nullptr, 1, 1};
      auto run = [&]() {
        return sgic_gemm_split3_f32(nullptr, 0, 0, 0, Ap, Wp, bias, sh.res ? R : nullptr, sh.N, C, sh.N, nullptr, sh.M, sh.N, sh.K, sh.act, 0, 0, &o, nullptr);
      };
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      // settle the clock: ~0.5 s of back-to-back launches
      hipEventRecord(e0);
      int reps = 0;
      float ms = 0;
      do {
        for (int i = 0; i < 50; i++) run();
        reps += 50;
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      } while (ms < 500.f);
      hipEventRecord(e0);
      for (int i = 0; i < 50; i++) run();
      hipEventRecord(e1);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 50;
      unsigned long long *dst;
      hipGetSymbolAddress((void **)&dst, HIP_SYMBOL(s3_wg));
      std::vector<unsigned long long> st(1024 * 6);
      hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
      // the last launch's workgroups (modes with two launches: the stamps of the tail launch overwrite low block ids -- skip 12)
      const bool two = mode == 12;
      int TM = 128, TN = (mode == 1 || mode == 10) ? 256 : 128;
      long tiles = (long)((sh.M + TM - 1) / TM) * ((sh.N + TN - 1) / TN);
      int nwg = (int)std::min<long>(tiles, (mode >= 10) ? 256 : 1024);
      double pro = 0, mn = 0, ep = 0, cyc = 0, rt = 0, nt = 0;
      unsigned long long first = ~0ull, last = 0, first_end = ~0ull;
      if (!two)
        for (int b = 0; b < nwg; b++) {
          const auto *s = &st[b * 6];
          pro += s[0], mn += s[1], ep += s[2], nt += s[3];
          cyc += s[0] + s[1] + s[2];
          rt += (double)(s[5] - s[4]);
          first = std::min(first, s[4]);
          last = std::max(last, s[5]);
          first_end = std::min(first_end, s[5]);
        }
      const int mf_per_slice = (TN == 256 ? 16 : 8) * 6;   // blocks per wave x 6
      const double floor_per_tile = (double)(sh.K / 32) * mf_per_slice * 16 * 2;
      printf("(%d,%d,%d,res=%d,act=%d) mode %2d: %.1f us %.1f TF (%.3f of 419.4)", sh.M, sh.N, sh.K, sh.res, sh.act, mode, ms * 1e3,
             2.0 * sh.M * sh.N * sh.K / ms / 1e9, 2.0 * sh.M * sh.N * sh.K / ms / 1e9 / 419.4);
      if (!two && nt > 0)
        printf(" | clock %.2f GHz | wg life: prologue %.1f%% main %.1f%% epilogue %.1f%% | main loop = %.3f of its MFMA floor, epilogue %.0f cyc/tile, prologue %.0f | "
               "tiles/wg %.2f | span %.1f us, first wg ends at %.1f us",
               cyc / rt / 10.0 * 1e-0, 100 * pro / cyc, 100 * mn / cyc, 100 * ep / cyc, floor_per_tile * nt / mn, ep / nt, pro / nt, nt / nwg,
               (double)(last - first) / 100.0, (double)(first_end - first) / 100.0);
      printf("\n");
      fflush(stdout);
    }
    hipFree(A), hipFree(W), hipFree(C), hipFree(R), hipFree(bias), hipFree(Ap), hipFree(Wp);
  }
  return 0;
}
