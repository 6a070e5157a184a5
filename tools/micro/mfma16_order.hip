// Is a v_mfma_f32_16x16x4_f32 chain bitwise equal to the v_mfma_f32_32x32x2_f32 chain of csrc/gemm.hip when the k's are fed in
// the same order?  gemm.hip's order inside an 8-k sub-step s: for t = 0..3: k = 8s+t (lanes 0-31), then k = 8s+4+t (lanes 32-63),
// i.e. 0,4,1,5,2,6,3,7.  A 16x16x4 MFMA takes 4 k's, one per 16-lane group g = lane >> 4, presumably accumulated g = 0..3:
// MFMA a = (8s+0, 8s+4, 8s+1, 8s+5), MFMA b = (8s+2, 8s+6, 8s+3, 8s+7).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_order tools/micro/mfma16_order.hip && /tmp/mfma16_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C[32][32] = A[32][K] . W[32][K]^T with the 32x32x2 chain (one wave)
__global__ void k32(const float *A, const float *W, float *C, int K) {
  const int lane = threadIdx.x, lrow = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  for (int s = 0; s < K / 8; s++) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(A + lrow * K + s * 8 + lh * 4);
    const f32x4 w = *reinterpret_cast<const f32x4 *>(W + lrow * K + s * 8 + lh * 4);
    for (int t = 0; t < 4; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], w[t], acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; e++) C[((e & 3) + 8 * (e >> 2) + 4 * lh) * 32 + lrow] = acc[e];
}

// the same C as four 16x16 blocks, one wave each (blockIdx = block), with the 16x16x4 chain
__global__ void k16(const float *A, const float *W, float *C, int K) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const int bi = blockIdx.x >> 1, bj = blockIdx.x & 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < K / 8; s++) {
    // group g holds the float4 of k-half (g & 1); groups 0,1 feed elements 0 / 2, groups 2,3 elements 1 / 3
    const f32x4 a = *reinterpret_cast<const f32x4 *>(A + (bi * 16 + r) * K + s * 8 + (g & 1) * 4);
    const f32x4 w = *reinterpret_cast<const f32x4 *>(W + (bj * 16 + r) * K + s * 8 + (g & 1) * 4);
    const int e0 = g >> 1;        // 0 for groups 0,1; 1 for groups 2,3
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e0 ? a[1] : a[0], e0 ? w[1] : w[0], acc, 0, 0, 0);   // k = 0,4,1,5
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(e0 ? a[3] : a[2], e0 ? w[3] : w[2], acc, 0, 0, 0);   // k = 2,6,3,7
  }
  // D layout of 16x16x4: lane (col = lane & 15, rows 4*(lane >> 4) + i)
  for (int i = 0; i < 4; i++) C[(bi * 16 + 4 * g + i) * 32 + bj * 16 + r] = acc[i];
}

int main() {
  const int K = 1024;
  float *hA = (float *)malloc(32 * K * 4), *hW = (float *)malloc(32 * K * 4);
  srand(1);
  for (int i = 0; i < 32 * K; i++) hA[i] = (float)rand() / RAND_MAX * 2 - 1, hW[i] = (float)rand() / RAND_MAX * 2 - 1;
  float *A, *W, *C1, *C2;
  hipMalloc(&A, 32 * K * 4); hipMalloc(&W, 32 * K * 4); hipMalloc(&C1, 4096); hipMalloc(&C2, 4096);
  hipMemcpy(A, hA, 32 * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(W, hW, 32 * K * 4, hipMemcpyHostToDevice);
  k32<<<1, 64>>>(A, W, C1, K);
  k16<<<4, 64>>>(A, W, C2, K);
  float h1[1024], h2[1024];
  hipMemcpy(h1, C1, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(h2, C2, 4096, hipMemcpyDeviceToHost);
  int diff = 0;
  double maxd = 0;
  for (int i = 0; i < 1024; i++) {
    if (memcmp(&h1[i], &h2[i], 4)) diff++;
    if (fabs((double)h1[i] - h2[i]) > maxd) maxd = fabs((double)h1[i] - h2[i]);
  }
  // fp64 reference of C[0][0] in the canonical order with fmaf
  float ref = 0.f;
  for (int s = 0; s < K / 8; s++) {
    const int ord[8] = {0, 4, 1, 5, 2, 6, 3, 7};
    for (int j = 0; j < 8; j++) ref = fmaf(hA[s * 8 + ord[j]], hW[s * 8 + ord[j]], ref);
  }
  printf("K=%d: %d of 1024 outputs differ bitwise between the 32x32x2 chain and the 16x16x4 chain (max |d| %.3g); C[0][0]: mfma32 %.9g mfma16 %.9g fmaf-chain %.9g\n",
         K, diff, maxd, h1[0], h2[0], ref);
  return 0;
}
