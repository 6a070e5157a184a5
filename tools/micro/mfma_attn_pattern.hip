// Microbenchmark: the MFMA pattern of the attention kernel in isolation -- 32 dependent MFMAs into S, then 32 MFMAs
// whose B operand is an element of the S accumulator (P consumed from the accumulator registers) into two O chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0) {
  f32x16 s, o0, o1;
  float qf[32];
#pragma unroll
  for (int e = 0; e < 16; e++) o0[e] = 0.f, o1[e] = 0.f;
#pragma unroll
  for (int i = 0; i < 32; i++) qf[i] = a0 + i + threadIdx.x;
  float a = a0 + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, qf[i], s, 0, 0, 0);
    if (MODE == 1) {  // exp on the accumulator between the phases (softmax-like VALU on S)
#pragma unroll
      for (int e = 0; e < 16; e++) s[e] = __expf(s[e] * 1e-30f);
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[e], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a + 1.f, s[e], o1, 0, 0, 0);
    }
  }
  float r = 0.f;
#pragma unroll
  for (int e = 0; e < 16; e++) r += o0[e] + o1[e];
  if (r == 12345.f) out[0] = r;
}

template <int MODE>
void run(int wgs_per_cu, float *d) {
  const int iters = 2000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(d, iters, 1.f);
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(d, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * iters * 64 * 4096.0;
  printf("mode %d (0 = MFMA only, 1 = + exp on S)  waves/SIMD %d : %.1f TFLOP/s\n", MODE, wgs_per_cu, fl / ms / 1e9);
}

int main() {
  float *d;
  hipMalloc(&d, 4);
  for (int w = 1; w <= 4; w++) {
    run<0>(w, d);
    run<1>(w, d);
  }
  return 0;
}
