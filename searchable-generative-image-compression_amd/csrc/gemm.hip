// fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// This is THE dominant kernel of the codec (>= 97 % of the FLOPs: every nn.Linear, 1x1 conv, the
// patch-embed / 2x2 convs as im2col GEMMs; reference call sites titok/blocks.py:37-64,
// blocks/swin_transformer.py:94-156, models/cross_blocks.py:75-98, blocks/conv_blocks.py:71-81,
// blocks/dcvc.py:28-54).  The reference computes in fp32, so we use the exact-fp32 MFMA
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 157 TFLOP/s chip peak); its result is bit-for-bit a
// k-ordered fmaf chain, and the k order here is fixed by K alone (no split-K), so every output row is
// independent of M and of the batch it sits in (batch-invariant: B=32 equals 32 x B=1).
//
// Tiling (wave64): 256 threads = 4 waves as 2(M) x 2(N); workgroup tile 128x128, wave tile 64x64 =
// 2x2 MFMA 32x32 blocks (64 accumulator VGPRs); BK = 32.  A and W tiles are staged global -> VGPR
// (float4, issued before the MFMA phase so HBM/L2 latency hides under 64 MFMAs per wave) -> LDS with
// a 36-float row stride (144 B: the 16 lanes of a ds_read_b128 group hit 16 distinct 16-B slots).
// Each lane fetches its operands with ds_read_b128 (4 consecutive k of one row); MFMA step t of a
// group uses element t of both fragments, i.e. lanes 0-31 feed k = 8s+t and lanes 32-63 k = 8s+4+t.
// 36 KB LDS and ~110 VGPRs per workgroup -> 2-3 workgroups per CU, so one workgroup's staging
// overlaps another's MFMAs.  blockIdx is remapped (bijectively) so that the workgroups sharing an
// XCD's L2 work on neighbouring tiles of the same A row-panels.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 32
#define LDS_LD 36

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2, ACT_TANH = 3, ACT_LRELU = 4 };

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    case ACT_SILU: return v / (1.0f + expf(-v));
    case ACT_TANH: return tanhf(v);
    case ACT_LRELU: return v >= 0.f ? v : 0.01f * v;
    default: return v;
  }
}

struct GemmArgs {
  const float *A;
  const float *W;
  const float *bias;  // [N] or null
  const float *R;     // residual [M, ldr] or null (added after the activation)
  float *C;
  int M, N, K;
  int lda, ldw, ldr, ldc;
  int act;
  int tiles_m, tiles_n;
  // optional row maps  row(m) = (m / seg) * seg_stride + (m % seg)  (seg == 0: identity) so a GEMM can
  // read / write the [:, a:b] token slice of an (n, L, C) buffer in place (models/cross_blocks.py:87-94)
  int a_seg, a_seg_stride, c_seg, c_seg_stride;
};

__device__ __forceinline__ f32x4 ld4_guard(const float *p, bool ok) {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return ok ? *reinterpret_cast<const f32x4 *>(p) : z;
}

__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float sB[BN * LDS_LD];

  // ---- XCD-aware bijective remap: hardware deals consecutive block ids round-robin over 8 XCDs ----
  const int nwg = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  // staging map: 128 rows x 8 float4 per tile = 1024 float4 -> 4 per thread per operand
  // thread t handles rows (t>>3) + 32*i, i=0..3, chunk (t&7)
  const int srow = tid >> 3, schunk = tid & 7;
  f32x4 ra[4], rb[4];

  auto issue_loads = [&](int k0) {
    const int kk = k0 + schunk * 4;
    const bool kok = kk < g.K;  // K % 4 == 0 is required, so a chunk is all-in or all-out
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = srow + 32 * i;
      const int am = m0 + r;
      const size_t arow = g.a_seg ? (size_t)(am / g.a_seg) * g.a_seg_stride + (am % g.a_seg) : (size_t)am;
      ra[i] = ld4_guard(g.A + arow * g.lda + kk, kok && am < g.M);
      rb[i] = ld4_guard(g.W + (size_t)(n0 + r) * g.ldw + kk, kok && (n0 + r) < g.N);
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = srow + 32 * i;
      *reinterpret_cast<f32x4 *>(&sA[r * LDS_LD + schunk * 4]) = ra[i];
      *reinterpret_cast<f32x4 *>(&sB[r * LDS_LD + schunk * 4]) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int lrow = lane & 31, lhalf = lane >> 5;
  const float *pa0 = &sA[(wm + lrow) * LDS_LD + lhalf * 4];
  const float *pa1 = pa0 + 32 * LDS_LD;
  const float *pb0 = &sB[(wn + lrow) * LDS_LD + lhalf * 4];
  const float *pb1 = pb0 + 32 * LDS_LD;

  const int nk = (g.K + BK - 1) / BK;
  issue_loads(0);
  for (int kt = 0; kt < nk; ++kt) {
    store_lds();
    __syncthreads();
    if (kt + 1 < nk) issue_loads((kt + 1) * BK);  // in flight during the MFMA phase below
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(pa0 + s * 8);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(pa1 + s * 8);
      const f32x4 b0 = *reinterpret_cast<const f32x4 *>(pb0 + s * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4 *>(pb1 + s * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: D[row = (e&3) + 8*(e>>2) + 4*lhalf][col = lrow] ----
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + wn + j * 32 + lrow;
    if (n >= g.N) continue;
    const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        if (m < g.M) {
          float v = apply_act(acc[i][j][e] + bv, g.act);
          if (g.R) v += g.R[(size_t)m * g.ldr + n];
          const size_t crow = g.c_seg ? (size_t)(m / g.c_seg) * g.c_seg_stride + (m % g.c_seg) : (size_t)m;
          g.C[crow * g.ldc + n] = v;
        }
      }
    }
  }
}

// C[M,N] = act(A[M,K] @ W[N,K]^T + bias) + R      (nn.Linear / 1x1 conv semantics, all fp32)
extern "C" int sgic_gemm_f32(const float *d_A, int lda, const float *d_W, int ldw, const float *d_bias,
                             const float *d_R, int ldr, float *d_C, int ldc, int M, int N, int K, int act,
                             int a_seg, int a_seg_stride, int c_seg, int c_seg_stride, sgic_stream_t stream) {
  SGIC_REQUIRE(d_A && d_W && d_C && M > 0 && N > 0 && K > 0, "null/empty");
  SGIC_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0, "K, lda, ldw must be multiples of 4 floats");
  SGIC_REQUIRE(lda >= K && ldw >= K && ldc >= N && (!d_R || ldr >= N), "leading dimensions");
  SGIC_REQUIRE(((uintptr_t)d_A & 15) == 0 && ((uintptr_t)d_W & 15) == 0, "A and W must be 16-byte aligned");
  SGIC_REQUIRE(act >= 0 && act <= ACT_LRELU, "activation");
  SGIC_REQUIRE(a_seg >= 0 && c_seg >= 0 && (a_seg == 0 || a_seg_stride >= a_seg) && (c_seg == 0 || c_seg_stride >= c_seg),
               "row segment maps");
  GemmArgs g{d_A, d_W, d_bias, d_R, d_C, M, N, K, lda, ldw, ldr, ldc, act, (M + BM - 1) / BM, (N + BN - 1) / BN,
             a_seg, a_seg_stride, c_seg, c_seg_stride};
  gemm_f32_kernel<<<g.tiles_m * g.tiles_n, 256, 0, to_stream(stream)>>>(g);
  return sgic::check_launch("gemm_f32_kernel");
}
