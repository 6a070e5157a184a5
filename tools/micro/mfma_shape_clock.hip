// Does the chip hold a different clock on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md, DVFS
// give-back item 7: for bf16 the 16x16 shape delivered 1.15x the FLOP/s of the 32x32 shape at equal cycles per FLOP)?
// Bare MFMA loops on random operands in registers, same 64x64 output tile per wave (64 accumulator registers), 1 and 2 waves
// per SIMD, every CU busy; reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/msc tools/micro/mfma_shape_clock.hip && /tmp/msc
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k32(float *out, const float *in, int iters, long long *clk) {
  f32x16 acc[4];
  for (int b = 0; b < 4; b++)
    for (int e = 0; e < 16; e++) acc[b][e] = 0.f;
  float a[4], w[4];
  for (int i = 0; i < 4; i++) a[i] = in[threadIdx.x + 256 * i], w[i] = in[threadIdx.x + 256 * (i + 4)];
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 8; s++) {   // 8 x (2x2 blocks) x 32x32x2  = 16 k of a 64x64 wave tile
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1], w[s & 1], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1], w[2 + (s & 1)], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 + (s & 1)], w[s & 1], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 + (s & 1)], w[2 + (s & 1)], acc[3], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0.f;
  for (int b = 0; b < 4; b++)
    for (int e = 0; e < 16; e++) r += acc[b][e];
  if (r == 12345.678f) out[0] = r;
  if (threadIdx.x == 0 && blockIdx.x < 1024) clk[blockIdx.x * 2] = t1 - t0, clk[blockIdx.x * 2 + 1] = r1 - r0;
}

__global__ __launch_bounds__(256) void k16(float *out, const float *in, int iters, long long *clk) {
  f32x4 acc[16];
  for (int b = 0; b < 16; b++)
    for (int e = 0; e < 4; e++) acc[b][e] = 0.f;
  float a[4], w[4];
  for (int i = 0; i < 4; i++) a[i] = in[threadIdx.x + 256 * i], w[i] = in[threadIdx.x + 256 * (i + 4)];
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 4; s++) {   // 4 x (4x4 blocks) x 16x16x4 = 16 k of a 64x64 wave tile
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + s) & 3], w[(j + s) & 3], acc[i * 4 + j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float r = 0.f;
  for (int b = 0; b < 16; b++)
    for (int e = 0; e < 4; e++) r += acc[b][e];
  if (r == 12345.678f) out[0] = r;
  if (threadIdx.x == 0 && blockIdx.x < 1024) clk[blockIdx.x * 2] = t1 - t0, clk[blockIdx.x * 2 + 1] = r1 - r0;
}

template <typename K>
void run(K kern, const char *name, int per_cu, float *out, float *in, long long *clk) {
  const int iters = 40000, grid = 256 * per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) kern<<<grid, 256>>>(out, in, iters, clk);   // warm: ~2 s of load before the timed launch
  hipEventRecord(e0);
  kern<<<grid, 256>>>(out, in, iters, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[2048];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double ghz = 0;
  int n = grid < 1024 ? grid : 1024;
  for (int i = 0; i < n; i++) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
  const double flops = (double)grid * 4 * iters * 16 * 2.0 * 64 * 64;
  printf("%s, %d workgroup(s)/CU: %.1f TFLOP/s, in-kernel clock %.3f GHz, %.1f ms\n", name, per_cu, flops / ms / 1e9, ghz / n, ms);
}

int main() {
  float *out, *in;
  long long *clk;
  hipMalloc(&out, 4);
  hipMalloc(&in, 2048 * 4);
  hipMalloc(&clk, 2048 * 8);
  float h[2048];
  for (int i = 0; i < 2048; i++) h[i] = (float)((i * 2654435761u >> 7) & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int per_cu = 1; per_cu <= 2; per_cu++) {
    run(k32, "v_mfma_f32_32x32x2_f32", per_cu, out, in, clk);
    run(k16, "v_mfma_f32_16x16x4_f32", per_cu, out, in, clk);
  }
  return 0;
}
