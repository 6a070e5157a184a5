// bf16x3 split of an fp32 value (see gemm_split.hip): x = x1 + x2 + x3 with three bf16 pieces, exact for every normal fp32
// whose low piece stays a normal bf16.  Shared by the kernels that write GEMM operands directly as planes.
#pragma once
#include <hip/hip_runtime.h>

typedef __bf16 s3_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned s3_u32x2 __attribute__((ext_vector_type(2)));
typedef float s3_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned s3_pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: round to nearest even
  s3_bf16x2 r = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, r);
}
// (x0, x1) -> the three bf16 pairs; the residuals are exact in fp32 (each step removes the leading 8 bits)
// contract(off): after inlining, the subtractions must not fuse with a multiply that produced x (the planes are the split of
// the ROUNDED fp32 value, so a producer that writes planes and one that writes fp32 + split pass agree bit for bit)
__device__ __forceinline__ void s3_split_pair(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3) {
#pragma clang fp contract(off)
  p1 = s3_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
  p2 = s3_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
  p3 = s3_pk_bf16(s0, s1);
}
// four consecutive values of a row -> 8 bytes in each of the three planes (dst = element offset of the first value)
__device__ __forceinline__ void s3_store4(unsigned short *planes, long plane_elems, size_t dst, const s3_f32x4 &v) {
  unsigned a0, b0, c0, a1, b1, c1;
  s3_split_pair(v[0], v[1], a0, b0, c0);
  s3_split_pair(v[2], v[3], a1, b1, c1);
  *reinterpret_cast<s3_u32x2 *>(planes + dst) = s3_u32x2{a0, a1};
  *reinterpret_cast<s3_u32x2 *>(planes + plane_elems + dst) = s3_u32x2{b0, b1};
  *reinterpret_cast<s3_u32x2 *>(planes + 2 * plane_elems + dst) = s3_u32x2{c0, c1};
}

// ---- slice-major ("packed") plane layout [3][cols / 32][rows][32]: the 64 bytes a row contributes to a 32-k slice sit next to the
// neighbouring rows' (16 rows of a slice = 1 KiB of consecutive bytes: whole cache lines for the GEMM's LDS-DMA pieces) instead of
// 2 cols bytes apart.  Every producer of an A operand (LayerNorm, attention, the GEMM epilogue, the split pass inside the GEMM entry
// point, the GroupNorm / halo-copy writers of the convolution's halo planes) writes it; the row-major layout [3][rows][cols] remains for
// C callers that split with sgic_split3_f32 (opts->a_packed = 0).
__device__ __forceinline__ size_t s3_pack_off(size_t row, int col, size_t rows) { return ((size_t)(col >> 5) * rows + row) * 32 + (col & 31); }
