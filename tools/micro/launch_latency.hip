// Diagnostic: what do the timestamps of ONE small dispatch contain?  (round-3 question: DESIGN quoted 5.3 us for the 64x64x512
// split GEMM from a back-to-back loop, the driver's box read 10.9-11.8 us from the dispatch's own event pair.)
// For the split GEMM's latency tile (mode 4: 32x32 tiles, 64-k stages) on one 64x64x512 product and three more B = 1 shapes,
// the same launches are timed four ways in one process:
//   loop      = (hipEventRecord ... N launches ... hipEventRecord) / N          -- throughput of back-to-back dispatches
//   pair b2b  = the (start, stop) event pair of hipExtLaunchKernel, launches back to back   (what bench.py's window records)
//   pair idle = the same pair with the queue drained and the host asleep 200 us before every launch (Python-paced callers)
//   in-kernel = s_memrealtime of the first workgroup's first instruction -> last workgroup's last store retired (S3_STAMPS)
// Run it under `rocprofv3 --kernel-trace` as well: the trace's duration column is the pair's twin.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/micro/bin/launch_latency tools/micro/launch_latency.hip
#define S3_STAMPS
#include <algorithm>
#include <stdarg.h>
#include <unistd.h>
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

extern "C" int sgic_profiler_create(sgic_profiler **out) {
  *out = new sgic_profiler();
  return 0;
}

int main() {
  struct Shape { int M, N, K, mode; } shapes[] = {{64, 64, 512, 4}, {289, 1024, 1024, 4}, {289, 1024, 4096, 4}, {9248, 1024, 1024, 11}};
  for (auto sh : shapes) {
    float *A, *W, *C;
    uint16_t *Ap, *Wp;
    hipMalloc(&A, (size_t)sh.M * sh.K * 4);
    hipMalloc(&W, (size_t)sh.N * sh.K * 4);
    hipMalloc(&C, (size_t)sh.M * sh.N * 4);
    hipMalloc(&Ap, (size_t)sh.M * sh.K * 6);
    hipMalloc(&Wp, (size_t)sh.N * sh.K * 6);
    std::vector<float> h((size_t)std::max(sh.M, sh.N) * sh.K);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(A, h.data(), (size_t)sh.M * sh.K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)sh.N * sh.K * 4, hipMemcpyHostToDevice);
    sgic_split3_f32(A, sh.K, sh.M, sh.K, 0, 0, Ap, nullptr);
    sgic_split3_f32(W, sh.K, sh.N, sh.K, 0, 0, Wp, nullptr);
    sgic_profiler prof;
    sgic_launch_opts plain{sh.mode, 0, nullptr}, timed{sh.mode, 0, &prof};
    auto run = [&](const sgic_launch_opts *o) {
      return sgic_gemm_split3_f32(nullptr, 0, 0, 0, Ap, Wp, nullptr, nullptr, 0, C, sh.N, nullptr, sh.M, sh.N, sh.K, 0, 0, 0, o, nullptr);
    };
    unsigned long long *dst;
    hipGetSymbolAddress((void **)&dst, HIP_SYMBOL(s3_stamps));
    auto reset_stamps = [&]() {
      const unsigned long long init[4] = {~0ull, 0ull, 0ull, 0ull};
      hipMemcpy(dst, init, sizeof(init), hipMemcpyHostToDevice);
    };
    auto read_stamps = [&](double &span_us, double &wg0_cycles) {
      unsigned long long v[4];
      hipMemcpy(v, dst, sizeof(v), hipMemcpyDeviceToHost);
      span_us = (double)(v[1] - v[0]) / 100.0;   // 100 MHz
      wg0_cycles = (double)(v[3] - v[2]);
    };
    for (int rep = 0; rep < 20; rep++) run(&plain);
    hipDeviceSynchronize();
    // 1. back-to-back loop
    const int N = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < N; rep++) run(&plain);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float loop_ms;
    hipEventElapsedTime(&loop_ms, e0, e1);
    // 2. event pairs, back to back
    auto pairs = [&](bool idle, std::vector<float> &out, std::vector<double> &spans, double &wg0c) {
      const int R = 16;
      prof.n = 0;
      out.clear();
      spans.clear();
      for (int rep = 0; rep < R; rep++) {
        if (idle) {
          hipDeviceSynchronize();
          usleep(200);
          reset_stamps();
          hipDeviceSynchronize();
          usleep(200);
        }
        run(&timed);
        if (idle) {
          hipDeviceSynchronize();
          double s, c;
          read_stamps(s, c);
          spans.push_back(s);
          wg0c = c;
        }
      }
      hipDeviceSynchronize();
      for (int i = 0; i < prof.n; i++) {
        float ms;
        hipEventElapsedTime(&ms, prof.pool[i].first, prof.pool[i].second);
        out.push_back(ms * 1e3f);
      }
      prof.n = -1;
    };
    std::vector<float> b2b, idl;
    std::vector<double> sp, sp2;
    double wg0c = 0, dummy;
    pairs(false, b2b, sp2, dummy);
    pairs(true, idl, sp, wg0c);
    auto stat = [](std::vector<float> v, float &mn, float &med) {
      std::sort(v.begin(), v.end());
      mn = v.front();
      med = v[v.size() / 2];
    };
    float bmin, bmed, imin, imed;
    stat(b2b, bmin, bmed);
    stat(idl, imin, imed);
    std::sort(sp.begin(), sp.end());
    printf("(%d,%d,%d) mode %d: loop %.2f us/launch | event pair back-to-back min %.2f med %.2f | event pair after idle min %.2f med %.2f | "
           "in-kernel span (first wave start -> last store retired) min %.2f med %.2f us, workgroup 0: %.0f shader cycles\n",
           sh.M, sh.N, sh.K, sh.mode, loop_ms * 1e3 / N, bmin, bmed, imin, imed, sp.front(), sp[sp.size() / 2], wg0c);
    hipFree(A);
    hipFree(W);
    hipFree(C);
    hipFree(Ap);
    hipFree(Wp);
  }
  return 0;
}
