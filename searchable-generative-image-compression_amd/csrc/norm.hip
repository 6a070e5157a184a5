// Row-wise LayerNorm (+ optional fused activation) for fp32 "token x channel" activations.
// Reference: every nn.LayerNorm on the path (titok/blocks.py:36,42; blocks/swin_transformer.py:135,142;
// blocks/conv_blocks.py:62; models/cross_blocks.py:62,67; models/codec_sq_fixbpp.py:92,412,418).
// HBM-bound: one wave per row, float4 loads, row kept in registers (C <= 2048, C % 256 == 0 fast path),
// two-pass mean / centred variance with wave64 shuffles, biased variance, eps inside the sqrt.
// Rows can be addressed through a segment map  row(m) = (m / seg) * seg_stride + (m % seg)  so a slice
// [:, a:b] of an (n, L, C) token buffer is normalised in place without a gather.
#include "common.h"
#include "split3.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float act_f(float v, int act) {
  if (act == 2) return v / (1.0f + expf(-v));  // SiLU
  return v;
}

// planes != null: the normalised row goes out as bf16x3 planes [3][M][C] (the A operand of a split GEMM, gemm_split.hip)
// instead of y
template <int NV>  // NV float4 per lane: C = NV*256
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, int ldx, int xseg, int xseg_stride,
                                                        const float *__restrict__ gamma, const float *__restrict__ beta,
                                                        float *__restrict__ y, int ldy, int yseg, int yseg_stride, int M,
                                                        int C, float eps, int act, unsigned short *__restrict__ planes) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long xr = xseg ? (long)(m / xseg) * xseg_stride + (m % xseg) : m;
  const long yr = yseg ? (long)(m / yseg) * yseg_stride + (m % yseg) : m;
  const float *xp = x + xr * ldx;
  float *yp = y + yr * ldy;
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    v[i] = *reinterpret_cast<const f32x4 *>(xp + (i * 64 + lane) * 4);
    s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; i++) {
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float d = v[i][t] - mean;
      q += d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int c = (i * 64 + lane) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4 *>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(beta + c);
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = act_f((v[i][t] - mean) * rstd * g[t] + b[t], act);
    if (planes) s3_store4(planes, (long)M * C, s3_pack_off(m, c, M), o);
    else *reinterpret_cast<f32x4 *>(yp + c) = o;
  }
}

// generic C (multiple of 4): re-reads the row from L1/L2 instead of holding it in registers
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float *__restrict__ x, int ldx, int xseg,
                                                                int xseg_stride, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float *__restrict__ y,
                                                                int ldy, int yseg, int yseg_stride, int M, int C,
                                                                float eps, int act) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const long xr = xseg ? (long)(m / xseg) * xseg_stride + (m % xseg) : m;
  const long yr = yseg ? (long)(m / yseg) * yseg_stride + (m % yseg) : m;
  const float *xp = x + xr * ldx;
  float *yp = y + yr * ldy;
  float s = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + c);
    s += (v[0] + v[1]) + (v[2] + v[3]);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + c);
#pragma unroll
    for (int t = 0; t < 4; t++) q += (v[t] - mean) * (v[t] - mean);
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  for (int c = lane * 4; c < C; c += 256) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(xp + c);
    f32x4 o;
#pragma unroll
    for (int t = 0; t < 4; t++) o[t] = act_f((v[t] - mean) * rstd * gamma[c + t] + beta[c + t], act);
    *reinterpret_cast<f32x4 *>(yp + c) = o;
  }
}

static int layernorm_any(const float *d_x, int ldx, int xseg, int xseg_stride, const float *d_gamma, const float *d_beta,
                         float *d_y, int ldy, int yseg, int yseg_stride, int M, int C, float eps, int act,
                         unsigned short *d_planes, sgic_stream_t stream) {
  SGIC_REQUIRE(d_x && d_gamma && d_beta && (d_y || d_planes) && M > 0 && C > 0, "args");
  SGIC_REQUIRE((C & 3) == 0 && (ldx & 3) == 0 && ldx >= C, "C, ldx multiples of 4");
  SGIC_REQUIRE(d_planes || ((ldy & 3) == 0 && ldy >= C), "ldy");
  SGIC_REQUIRE(!d_planes || (C % 256 == 0 && C <= 2048 && ((uintptr_t)d_planes & 7) == 0), "planes output: C % 256 == 0, C <= 2048");
  SGIC_REQUIRE((((uintptr_t)d_x | (uintptr_t)d_y | (uintptr_t)d_gamma | (uintptr_t)d_beta) & 15) == 0, "alignment");
  SGIC_REQUIRE(act == 0 || act == 2, "act: none or SiLU");
  const unsigned grid = cdiv(M, 4);
  hipStream_t st = to_stream(stream);
#define LN_LAUNCH(NV)                                                                                              \
  layernorm_kernel<NV><<<grid, 256, 0, st>>>(d_x, ldx, xseg, xseg_stride, d_gamma, d_beta, d_y, ldy, yseg, yseg_stride, \
                                              M, C, eps, act, d_planes)
  if (C % 256 == 0 && C <= 2048) {
    switch (C / 256) {
      case 1: LN_LAUNCH(1); break;
      case 2: LN_LAUNCH(2); break;
      case 3: LN_LAUNCH(3); break;
      case 4: LN_LAUNCH(4); break;
      case 5: LN_LAUNCH(5); break;
      case 6: LN_LAUNCH(6); break;
      case 7: LN_LAUNCH(7); break;
      default: LN_LAUNCH(8); break;
    }
  } else {
    layernorm_generic_kernel<<<grid, 256, 0, st>>>(d_x, ldx, xseg, xseg_stride, d_gamma, d_beta, d_y, ldy, yseg,
                                                   yseg_stride, M, C, eps, act);
  }
  return sgic::check_launch("layernorm_kernel");
}

extern "C" int sgic_layernorm_f32(const float *d_x, int ldx, int xseg, int xseg_stride, const float *d_gamma,
                                  const float *d_beta, float *d_y, int ldy, int yseg, int yseg_stride, int M, int C,
                                  float eps, int act, sgic_stream_t stream) {
  return layernorm_any(d_x, ldx, xseg, xseg_stride, d_gamma, d_beta, d_y, ldy, yseg, yseg_stride, M, C, eps, act, nullptr, stream);
}

// LayerNorm whose output is consumed only by a split GEMM: written directly as bf16x3 planes [3][M][C] (dense rows)
extern "C" int sgic_layernorm_split3_f32(const float *d_x, int ldx, int xseg, int xseg_stride, const float *d_gamma,
                                         const float *d_beta, uint16_t *d_planes, int M, int C, float eps, int act,
                                         sgic_stream_t stream) {
  SGIC_REQUIRE(d_planes, "planes");
  return layernorm_any(d_x, ldx, xseg, xseg_stride, d_gamma, d_beta, nullptr, 0, 0, 0, M, C, eps, act, d_planes, stream);
}
