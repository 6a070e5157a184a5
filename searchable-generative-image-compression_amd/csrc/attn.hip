// fp32 multi-head attention for head_dim 64 on the CDNA4 matrix cores (flash style, online softmax).
//
// Serves every attention on the path, all of which have 64-wide heads:
//   * TiTok ViT-L blocks, L=289, 16 heads            (titok/blocks.py:50-54 nn.MultiheadAttention)
//   * cross blocks, L=545, 12 heads                   (models/cross_blocks.py:88 via ResidualAttentionBlock)
//   * Swin window attention, L=256 per 16x16 window, dense additive bias (relative position table +
//     the -inf shift masks), cyclic shift folded into a row map   (blocks/swin_transformer.py:94-128)
//   * CLIP ViT-B/32, L=50, 12 heads                   (open_clip image tower, compress.py:72)
//
// One workgroup = 4 waves = 128 query rows of one (sequence, head); each wave owns 32 query rows.
// K/V tiles of 32 keys are staged in LDS by the whole workgroup.  Per tile and wave:
//   S^T = K . Q^T      32 x v_mfma_f32_32x32x2_f32   (A = K tile from LDS, B = Q^T kept in 32 VGPRs)
//   online softmax     the accumulator layout puts ONE query row on each lane (col = lane&31) with 16
//                      of its 32 keys in registers, so max/sum are 15 in-register ops + one
//                      cross-half shuffle
//   O^T += V^T . P^T   32 x MFMA; P^T is consumed straight from the S^T accumulator registers (the MFMA
//                      k index only has to pair the same key on both operands), V^T from LDS
// q is pre-scaled by `scale` (0.125 is a power of two, so this equals scaling the scores).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AT_KT 32      // keys per tile
#define AT_LDK 68     // padded row stride (floats) of the K tile: 272 B -> conflict-free ds_read_b128
#define AT_LDV 64

struct AttnArgs {
  const float *q, *k, *v;  // row-strided, head h at column offset h*64
  float *out;
  int ldq, ldk, ldv, ldo;
  int L, nseq, nheads;
  const int *rowmap;    // [nseq*L] row of (seq, token) or null => seq*L + token
  const float *bias;    // [nvar][L][L] additive bias or null
  const int *biasvar;   // [nseq] variant index or null (=> variant 0)
  float scale;
  int nwaves, qblocks;
};

// blockDim.x = 64 * a.nwaves; one workgroup covers 32*nwaves query rows of one (sequence, head); qblocks
// workgroups cover the sequence.  The host picks nwaves so that padding waste is small (L=289 -> 1 x 10 waves,
// L=545 -> 2 x 9, L=256 -> 1 x 8) and K/V are staged once per (sequence, head) where possible.
__global__ __launch_bounds__(640) void attn_f32_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float sK[AT_KT * AT_LDK];
  __shared__ __attribute__((aligned(16))) float sV[AT_KT * AT_LDV];

  const int qblocks = a.qblocks, nthreads = blockDim.x;
  int bid = blockIdx.x;
  const int qb = bid % qblocks;
  bid /= qblocks;
  const int head = bid % a.nheads;
  const int seq = bid / a.nheads;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = lane & 31, lh = lane >> 5;
  const int q_tok = (qb * a.nwaves + wave) * 32 + lq;  // this lane's query token
  const bool q_ok = q_tok < a.L;
  const bool wave_active = (qb * a.nwaves + wave) * 32 < a.L;
  const int q_tok_c = q_ok ? q_tok : a.L - 1;
  const long q_row = a.rowmap ? a.rowmap[(long)seq * a.L + q_tok_c] : (long)seq * a.L + q_tok_c;
  const int hc = head * 64;

  // Q^T fragment: lane (q, h) holds Q[q][(2c+h)*4 + t], c=0..7, t=0..3  (pairs with the K read below)
  float qf[32];
  {
    const float *qp = a.q + q_row * a.ldq + hc;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const f32x4 v4 = *reinterpret_cast<const f32x4 *>(qp + (2 * c + lh) * 4);
#pragma unroll
      for (int t = 0; t < 4; t++) qf[c * 4 + t] = v4[t] * a.scale;
    }
  }

  f32x16 o0, o1;  // O^T: rows d (0..31 / 32..63), col = query
#pragma unroll
  for (int e = 0; e < 16; e++) o0[e] = 0.f, o1[e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // l_run: this lane's partial row sum (its 16 keys per tile)

  const float *bias_base = nullptr;
  if (a.bias) {
    const int var = a.biasvar ? a.biasvar[seq] : 0;
    bias_base = a.bias + ((long)var * a.L + q_tok_c) * a.L;
  }

  // K/V staging: 32 keys x 16 float4 per operand = 512 float4 each; thread t stages float4 index t (and
  // t + nthreads if needed).  Loads are unconditional: keys past L are clamped to the last token -- their
  // scores are forced to -inf below, so P = 0 and the (finite) clamped K/V values never contribute.
  const int nst = (512 + nthreads - 1) / nthreads;  // 1 or 2
  f32x4 rk[2], rv[2];
  auto issue_stage = [&](int key0) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (i < nst) {
        const int f = tid + nthreads * i;
        if (f < 512) {
          const int kr = f >> 4, c4 = f & 15;
          const int tok = min(key0 + kr, a.L - 1);
          const long row = a.rowmap ? a.rowmap[(long)seq * a.L + tok] : (long)seq * a.L + tok;
          rk[i] = *reinterpret_cast<const f32x4 *>(a.k + row * a.ldk + hc + c4 * 4);
          rv[i] = *reinterpret_cast<const f32x4 *>(a.v + row * a.ldv + hc + c4 * 4);
        }
      }
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if (i < nst) {
        const int f = tid + nthreads * i;
        if (f < 512) {
          const int kr = f >> 4, c4 = f & 15;
          *reinterpret_cast<f32x4 *>(&sK[kr * AT_LDK + c4 * 4]) = rk[i];
          *reinterpret_cast<f32x4 *>(&sV[kr * AT_LDV + c4 * 4]) = rv[i];
        }
      }
    }
  };

  // Ragged tail: L = 289 and 545 are 9 / 17 full key tiles plus ONE key.  A 32-key MFMA tile for one key is
  // 97 % padding, so a tail of <= 2 keys is folded into the online softmax on the VALU instead (a rank-1 update).
  const int rem = a.L % AT_KT;
  const bool tail_valu = rem > 0 && rem <= 2;
  const int ntiles = tail_valu ? a.L / AT_KT : (a.L + AT_KT - 1) / AT_KT;
  issue_stage(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    const int key0 = kt * AT_KT;
    __syncthreads();  // previous tile fully consumed
    store_stage();
    __syncthreads();
    if (kt + 1 < ntiles) issue_stage(key0 + AT_KT);  // next tile's loads fly during this tile's 64 MFMAs
    if (!wave_active) continue;  // a wave whose 32 query rows are all padding only helps staging (wave-uniform)

    // ---- S^T[key][q] = sum_d K[key][d] * Q[q][d] ----
    f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const f32x4 kf = *reinterpret_cast<const f32x4 *>(&sK[lq * AT_LDK + (2 * c + lh) * 4]);
#pragma unroll
      for (int t = 0; t < 4; t++) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[t], qf[c * 4 + t], s, 0, 0, 0);
    }
    // lane holds keys key0 + (e&3) + 8*(e>>2) + 4*lh  for its query
    if (bias_base) {
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int kb = min(key0 + 8 * g4 + 4 * lh, a.L - 4);  // L % 4 == 0 with a bias; clamped reads are masked below
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_base + kb);
#pragma unroll
        for (int t = 0; t < 4; t++) s[g4 * 4 + t] += b4[t];
      }
    }
    if (key0 + AT_KT > a.L) {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int key = key0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (key >= a.L) s[e] = -INFINITY;
      }
    }
    float mx = s[0];
#pragma unroll
    for (int e = 1; e < 16; e++) mx = fmaxf(mx, s[e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // whole row masked so far
    const float alpha = __expf(m_run - m_use);               // m_run = -inf -> 0 (v_exp_f32 path: ~3e-6 rel. at |x| = 50)
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; e++) {
      s[e] = __expf(s[e] - m_use);
      psum += s[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int e = 0; e < 16; e++) o0[e] *= alpha, o1[e] *= alpha;

    // ---- O^T[d][q] += sum_key V[key][d] * P[q][key];  k-step e pairs key (e&3)+8*(e>>2)+4*lh ----
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * lh;
      const float v0 = sV[key * AT_LDV + lq], v1 = sV[key * AT_LDV + 32 + lq];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[e], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[e], o1, 0, 0, 0);
    }
  }

  if (tail_valu && wave_active) {
    for (int key = ntiles * AT_KT; key < a.L; ++key) {
      const long row = a.rowmap ? a.rowmap[(long)seq * a.L + key] : (long)seq * a.L + key;
      const float *kp = a.k + row * a.ldk + hc, *vp = a.v + row * a.ldv + hc;
      float sc = 0.f;  // this lane's half of q . k (dims (2c+lh)*4 + t), completed by the cross-half shuffle
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const f32x4 k4 = *reinterpret_cast<const f32x4 *>(kp + (2 * c + lh) * 4);
#pragma unroll
        for (int t = 0; t < 4; t++) sc = fmaf(k4[t], qf[c * 4 + t], sc);
      }
      sc += __shfl_xor(sc, 32);
      if (bias_base) sc += bias_base[key];
      const float m_new = fmaxf(m_run, sc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __expf(m_run - m_use);
      const float pk = __expf(sc - m_use);
      l_run = l_run * alpha + (lh == 0 ? pk : 0.f);  // l_run is a per-half partial sum: count the key once
      m_run = m_new;
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(vp + 8 * g4 + 4 * lh);
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(vp + 32 + 8 * g4 + 4 * lh);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          o0[g4 * 4 + t] = fmaf(pk, v0[t], o0[g4 * 4 + t] * alpha);
          o1[g4 * 4 + t] = fmaf(pk, v1[t], o1[g4 * 4 + t] * alpha);
        }
      }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
    float *op = a.out + q_row * a.ldo + hc;
#pragma unroll
    for (int g4 = 0; g4 < 4; g4++) {
      const int d = 8 * g4 + 4 * lh;
      f32x4 w0, w1;
#pragma unroll
      for (int t = 0; t < 4; t++) w0[t] = o0[g4 * 4 + t] * inv, w1[t] = o1[g4 * 4 + t] * inv;
      *reinterpret_cast<f32x4 *>(op + d) = w0;
      *reinterpret_cast<f32x4 *>(op + 32 + d) = w1;
    }
  }
}

static int g_attn_max_waves = 10;
extern "C" int sgic_attention_set_max_waves(int w) {
  if (w < 4 || w > 10) return SGIC_EINVAL;
  g_attn_max_waves = w;
  return SGIC_OK;
}

extern "C" int sgic_attention_f32(const float *d_q, int ldq, const float *d_k, int ldk, const float *d_v, int ldv,
                                  float *d_out, int ldo, int L, int nseq, int nheads, const int32_t *d_rowmap,
                                  const float *d_bias, const int32_t *d_biasvar, float scale, sgic_stream_t stream) {
  SGIC_REQUIRE(d_q && d_k && d_v && d_out && L > 0 && nseq > 0 && nheads > 0, "args");
  SGIC_REQUIRE((ldq & 3) == 0 && (ldk & 3) == 0 && (ldv & 3) == 0 && (ldo & 3) == 0, "row strides must be multiples of 4");
  SGIC_REQUIRE(ldq >= nheads * 64 && ldk >= nheads * 64 && ldv >= nheads * 64 && ldo >= nheads * 64, "head_dim is 64");
  SGIC_REQUIRE((((uintptr_t)d_q | (uintptr_t)d_k | (uintptr_t)d_v | (uintptr_t)d_out) & 15) == 0, "16-byte alignment");
  SGIC_REQUIRE(!d_bias || (L & 3) == 0, "bias needs L % 4 == 0");
  const int rows32 = (L + 31) / 32;             // 32-row wave slices needed
  const int mw = g_attn_max_waves;
  const int qblocks = (rows32 + mw - 1) / mw;   // at most 10 waves (640 threads) per workgroup
  int nwaves = (rows32 + qblocks - 1) / qblocks;
  if (nwaves < 4) nwaves = 4;                   // >= 256 threads so the 2-slot K/V staging covers a tile
  AttnArgs a{d_q, d_k, d_v, d_out, ldq, ldk, ldv, ldo, L, nseq, nheads, d_rowmap, d_bias, d_biasvar, scale, nwaves, qblocks};
  const long grid = (long)nseq * nheads * qblocks;
  attn_f32_kernel<<<(unsigned)grid, 64 * nwaves, 0, to_stream(stream)>>>(a);
  return sgic::check_launch("attn_f32_kernel");
}
