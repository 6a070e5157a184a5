// Diagnostic: where does an attention launch spend its time?  Builds the product kernel (csrc/attn.hip) with four
// s_memtime stamps per wave and prints, per shape and attn_mode, the launch span and the per-wave phase medians.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/attn_stamps tools/micro/attn_stamps.hip && /tmp/attn_stamps
#define AT_STAMPS
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
#include "../../searchable-generative-image-compression_amd/csrc/attn.hip"

static double med(std::vector<double> &v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[v.size() / 2];
}

int main() {
  struct Shape { int L, nseq, heads, bias; } shapes[] = {{289, 32, 16, 0}, {545, 32, 12, 0}, {256, 32, 12, 1}};
  for (auto sh : shapes) {
    const int D = sh.heads * 64;
    const size_t rows = (size_t)sh.nseq * sh.L;
    float *qkv, *out, *bias = nullptr;
    hipMalloc(&qkv, rows * 3 * D * 4);
    hipMalloc(&out, rows * D * 4);
    std::vector<float> h(rows * 3 * D);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    if (sh.bias) {
      hipMalloc(&bias, (size_t)sh.L * sh.L * 4);
      hipMemset(bias, 0, (size_t)sh.L * sh.L * 4);
    }
    for (int mode : {1, 5, 9, 10, 13}) {   // 9 / 10 / 13 = 1 / 2 / 5 with both products as bf16x3 split MFMAs (the default arithmetic)
      sgic_launch_opts o{0, mode, nullptr};
      for (int rep = 0; rep < 3; rep++)
        sgic_attention_f32(qkv, 3 * D, qkv + D, 3 * D, qkv + 2 * D, 3 * D, out, D, sh.L, sh.nseq, sh.heads, nullptr, bias, nullptr,
                           0.125f, &o, nullptr);
      hipDeviceSynchronize();
      long long *dst;
      hipGetSymbolAddress((void **)&dst, HIP_SYMBOL(at_stamps));
      hipMemset(dst, 0, sizeof(long long) * 4 * (1 << 17));
      { long long *dr; hipGetSymbolAddress((void **)&dr, HIP_SYMBOL(at_real)); hipMemset(dr, 0, sizeof(long long) * 2 * (1 << 17)); }
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      sgic_attention_f32(qkv, 3 * D, qkv + D, 3 * D, qkv + 2 * D, 3 * D, out, D, sh.L, sh.nseq, sh.heads, nullptr, bias, nullptr,
                         0.125f, &o, nullptr);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> st(4 * (1 << 17));
      hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
      long long *dreal;
      hipGetSymbolAddress((void **)&dreal, HIP_SYMBOL(at_real));
      std::vector<long long> rl(2 * (1 << 17));
      hipMemcpy(rl.data(), dreal, rl.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> pro, loop, epi, start, endt;
      long long tmin = -1, tmax = 0;
      for (size_t w = 0; w < (1 << 17); w++) {
        if (!st[w * 4] || !st[w * 4 + 3]) continue;
        if (tmin < 0 || rl[w * 2] < tmin) tmin = rl[w * 2];
        tmax = std::max(tmax, rl[w * 2 + 1]);
      }
      for (size_t w = 0; w < (1 << 17); w++) {
        const long long *s = &st[w * 4];
        if (!s[0] || !s[3]) continue;
        start.push_back((double)(rl[w * 2] - tmin) / 100.0);      // microseconds
        endt.push_back((double)(rl[w * 2 + 1] - tmin) / 100.0);
        pro.push_back((double)(s[1] - s[0]));
        loop.push_back((double)(s[2] - s[1]));
        epi.push_back((double)(s[3] - s[2]));
      }
      const size_t nw = pro.size();
      std::sort(start.begin(), start.end());
      std::sort(endt.begin(), endt.end());
      printf("L=%d mode %d: %.1f us by events; first wave start -> last wave end %.1f us; waves %zu; start p10 %.1f p50 %.1f p90 %.1f max %.1f us | "
             "end p10 %.1f p50 %.1f p90 %.1f max %.1f us | cycles: prologue p50 %.0f | loop p10 %.0f p50 %.0f p90 %.0f max %.0f (p50 per 32-key tile %.0f; MFMA floor per tile and wave %d) | epilogue p50 %.0f\n",
             sh.L, mode, ms * 1e3, (tmax - tmin) / 100.0, nw, start[nw / 10], start[nw / 2], start[nw * 9 / 10], start[nw - 1],
             endt[nw / 10], endt[nw / 2], endt[nw * 9 / 10], endt[nw - 1], med(pro), (std::sort(loop.begin(), loop.end()), loop[nw / 10]), loop[nw / 2],
             loop[nw * 9 / 10], loop[nw - 1], loop[nw / 2] / (sh.L / 32), mode >= 8 ? 1536 : 4096, med(epi));
    }
    hipFree(qkv);
    hipFree(out);
    if (bias) hipFree(bias);
  }
  return 0;
}
