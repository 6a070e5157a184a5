// Microbenchmark: fp32 MFMA 32x32x2 throughput vs number of independent accumulator chains per wave and waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip ; run: ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[c][e] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 64 / CH; r++)
#pragma unroll
      for (int c = 0; c < CH; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int e = 0; e < 16; e++) s += acc[c][e];
  if (s == 12345.f) out[0] = s;
}

template <int CH>
void run(int wgs_per_cu, float *d) {
  const int iters = 2000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<CH><<<grid, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e0);
  k<CH><<<grid, 256>>>(d, iters, 1.f, 2.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 64 * 4096.0;
  printf("chains/wave %d  waves/SIMD %d : %.1f TFLOP/s\n", CH, wgs_per_cu, fl / ms / 1e9);
}

int main() {
  float *d;
  hipMalloc(&d, 4);
  for (int w = 1; w <= 4; w++) {
    run<1>(w, d);
    run<2>(w, d);
    run<4>(w, d);
  }
  return 0;
}
