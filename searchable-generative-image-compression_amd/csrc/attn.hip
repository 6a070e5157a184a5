// fp32 multi-head attention for head_dim 64 on the CDNA4 matrix cores (flash style, online softmax).
//
// Serves every attention on the path, all of which have 64-wide heads:
//   * TiTok ViT-L blocks, L=289, 16 heads            (titok/blocks.py:50-54 nn.MultiheadAttention)
//   * cross blocks, L=545, 12 heads                   (models/cross_blocks.py:88 via ResidualAttentionBlock)
//   * Swin window attention, L=256 per 16x16 window, dense additive bias (relative position table +
//     the -inf shift masks), cyclic shift folded into a row map   (blocks/swin_transformer.py:94-128)
//   * CLIP ViT-B/32 image tower L=50 / text tower L=77 (+causal bias)   (open_clip, compress.py:72, search.py:93-97)
//
// Work decomposition (round 2).  A *unit* is one (sequence, head); its queries are cut into 32-row blocks and the
// (unit, row block) pairs of the whole launch form ONE flat item list.  A workgroup = 4 waves = 4 CONSECUTIVE items,
// one per wave, so every workgroup carries four waves of identical work whatever L is (L = 289 and 545 are 9 and 17
// row blocks -- not multiples of anything; a workgroup-per-unit mapping leaves SIMDs idle or pads with empty waves).
// Four consecutive items touch at most two units; the K/V tiles of both are staged (LDS region 0 / 1) and each wave
// reads the region of its own unit.  Per 32-key tile and wave:
//   S^T = K . Q^T      32 x v_mfma_f32_32x32x2_f32   (A = K tile from LDS, B = Q^T kept in 32 VGPRs), split over two
//                      accumulators (dims 0-31 / 32-63) so consecutive MFMAs never wait on their own result
//   online softmax     the accumulator layout puts ONE query row on each lane (col = lane&31) with 16 of its 32
//                      keys in registers, so max/sum are 15 in-register ops + one cross-half shuffle
//   O^T += V^T . P^T   32 x MFMA; P^T is consumed straight from the S^T accumulator registers (the MFMA k index only
//                      has to pair the same key on both operands), V^T from LDS
// Staging is register-staged (global -> VGPR -> LDS) with a double-buffered LDS ring and ONE barrier per tile
// (NBUF = 2), or a single buffer with two barriers (NBUF = 1, half the LDS, more workgroups per CU); the loads of the
// next tile are in flight during the current tile's MFMAs, and with two units the second unit's loads reuse the same
// registers half a tile later.
// Ragged shapes: L % 32 in {1, 2} (289, 545) leaves one or two extra KEYS, folded in as a rank-1 VALU update instead of
// a 97 %-padding MFMA tile, and one or two extra QUERY rows, which are computed by pure-VALU workgroups (the first
// blocks of the same launch: they run beside the MFMA workgroups on the otherwise idle vector pipe) instead of a
// whole padded 32-row MFMA item per unit.
// Placement: workgroups that share units are dealt to the SAME XCD (blocks b and b+8 share an L2) in consecutive
// dispatch slots, so K/V of a unit is fetched from HBM once and re-read from that XCD's L2.
// q is pre-scaled by `scale` (0.125 is a power of two, so this equals scaling the scores).
#include "common.h"
#include "split3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AT_KT 32      // keys per tile
#define AT_LOG2E 1.4426950408889634f
// S3 variant: the running maximum is only raised (and O / l rescaled) when some row's maximum grew by more than this many powers of
// two; until then P = exp2(s - m_run) may exceed 1 (up to 2^AT_DEFER), which costs nothing in accuracy here -- P is split into three
// bf16 pieces EXACTLY and accumulated in fp32.  The decision is one __any() per tile over the wave's 32 rows, all of ONE (unit, row
// block) item, so a row's arithmetic depends on its item only: batch-invariant like everything else.
#define AT_DEFER 6.0f
#define AT_LDK 68     // padded row stride (floats) of the K tile: 272 B -> conflict-free ds_read_b128
#define AT_LDV 64
#define AT_TILE (AT_KT * AT_LDK + AT_KT * AT_LDV)   // floats of one unit's K+V tile (16.5 KiB)
// S3 variant (attn_mode bit 3): S^T = K . Q^T as a bf16x3 split product on v_mfma_f32_32x32x16_bf16 (gemm_split.hip's
// arithmetic: three bf16 pieces per fp32 operand, six MFMAs per 16 d, fp32 accumulate -- fp32-accurate, 6/16 of the matrix-pipe
// time of the fp32 MFMA).  K is split while it is staged (a tile is shared by the workgroup's four waves: 8 elements per
// thread), Q once per item into registers; the K tile lives in LDS as three bf16 planes [32 keys][64 d] with a 144-byte row
// stride (conflict-free ds_read_b128).  The S^T accumulator layout is that of the fp32 MFMA, so the softmax and
// O^T += V^T . P^T (fp32 MFMA, P straight from the accumulators) are unchanged.
#define AT_K3_ROWB 144                                   // bytes per key row of a K plane: 128 data + 16 pad
#define AT_K3_FLOATS (3 * AT_KT * AT_K3_ROWB / 4)        // the three K planes, in floats (13.5 KiB)
// ... and O^T += V^T . P^T the same way: V^T is staged as three bf16 planes [64 d][32 keys] (80-byte rows), the keys of a tile in
// the order the S^T accumulators hold them -- MFMA k-slot (h, j) of 16-key group G is key 16 G + 4 h + (j & 3) + 8 (j >> 2) --
// so a lane's eight P values of a group are its accumulator elements 8 G .. 8 G + 7 (split in registers after the exp) and its
// eight V values one 16-byte LDS read.  A staging thread owns one d and one (G, h) octet of keys: eight coalesced 4-byte loads
// (one key row per wave instruction), one ds_write_b128 per plane.
#define AT_V3_ROWB 80                                    // bytes per d row of a V^T plane: 64 data + 16 pad (conflict-free b128)
#define AT_V3_FLOATS (3 * 64 * AT_V3_ROWB / 4)           // the three V^T planes, in floats (15 KiB)
typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned at_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned at_u32x2 __attribute__((ext_vector_type(2)));
#define AT_RAG_MAXL 960                              // ragged-row path: q[64] + p[L] live in a wave-private 4 KiB LDS slice
#define AT_RAG_SLICE (64 + AT_RAG_MAXL)

// Diagnostic build only (tools/micro/attn_stamps.hip defines AT_STAMPS): four s_memtime stamps per wave -- start, main
// loop entered, main loop left, stores issued -- into a buffer nothing else reads.  The product build has no stamps.
#ifdef AT_STAMPS
__device__ long long at_stamps[4 * (1 << 17)];   // s_memtime (shader cycles; the counter base differs between dies)
__device__ long long at_real[2 * (1 << 17)];     // s_memrealtime at stamps 0 and 3 (100 MHz, one base for the chip)
#define AT_STAMP(i)                                                                                   \
  do {                                                                                                \
    if ((threadIdx.x & 63) == 0) {                                                                    \
      const size_t w_ = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);                                  \
      at_stamps[w_ * 4 + (i)] = (long long)__builtin_amdgcn_s_memtime();                              \
      if ((i) == 0 || (i) == 3) at_real[w_ * 2 + ((i) == 3)] = (long long)__builtin_amdgcn_s_memrealtime(); \
    }                                                                                                 \
  } while (0)
#else
#define AT_STAMP(i)
#endif

struct AttnArgs {
  const float *q, *k, *v;  // row-strided, head h at column offset h*64
  float *out;
  int ldq, ldk, ldv, ldo;
  int L, nseq, nheads;
  const int *rowmap;    // [nseq*L] row of (seq, token) or null => seq*L + token
  const float *bias;    // [nvar][L][L] additive bias or null
  const int *biasvar;   // [nseq] variant index or null (=> variant 0)
  float scale;
  int nq;        // 32-row query blocks per unit handled on the matrix cores
  int ipw;       // items per workgroup (4, or 2 when nq == 1 so that a workgroup never spans more than 2 units)
  int nqp;       // row-block slots per unit, >= nq (nq padded to a multiple of ipw: a workgroup then never spans two units)
  int n_items;   // nseq * nheads * nqp
  int n_wgs;     // MFMA workgroups (logical)
  int group;     // consecutive logical workgroups dealt to one XCD
  int rag;       // ragged query rows per unit computed on the VALU (0, 1 or 2): tokens nq*32 .. L-1
  int n_rag_wgs; // leading blocks that run the ragged-row path (4 rows per block)
  int stagger_cycles, first_round;  // start-up stagger of co-resident workgroups (see attn_f32_kernel), 0 = off
  unsigned short *out_planes;       // optional: the output as slice-major bf16x3 planes [3][nheads*2][rows][32] (operand of the out-projection
  long plane_elems;                 // split GEMM, gemm_split.hip) instead of `out`; rows * nheads * 64
};

__device__ __forceinline__ long at_row(const AttnArgs &a, int seq, int tok) {
  return a.rowmap ? (long)a.rowmap[(long)seq * a.L + tok] : (long)seq * a.L + tok;
}

// One ragged query row on the vector pipe: scores for all L keys (lane = key mod 64), softmax across the wave, then
// out[d = lane] = sum_key p[key] V[key][d].  K/V rows come straight from global memory (L2-hot: the MFMA workgroups of
// the same unit stream them at the same time); q and p live in a wave-private LDS slice.
__device__ void attn_ragged_row(const AttnArgs &a, int unit, int tok, float *wl /* AT_RAG_SLICE floats */, int lane) {
  const int head = unit % a.nheads, seq = unit / a.nheads, hc = head * 64;
  const long q_row = at_row(a, seq, tok);
  float *ql = wl, *pl = wl + 64;
  ql[lane] = a.q[q_row * a.ldq + hc + lane] * a.scale;
  const float *bias_row = nullptr;
  if (a.bias) bias_row = a.bias + ((long)(a.biasvar ? a.biasvar[seq] : 0) * a.L + tok) * a.L;
  float mx = -INFINITY;
  for (int key = lane; key < a.L; key += 64) {
    const float *kp = a.k + at_row(a, seq, key) * a.ldk + hc;
    float sc = 0.f;
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
      f32x4 k4[4];
#pragma unroll
      for (int j = 0; j < 4; j++) k4[j] = *reinterpret_cast<const f32x4 *>(kp + (c + j) * 4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 q4 = *reinterpret_cast<const f32x4 *>(ql + (c + j) * 4);   // same address on every lane: broadcast
#pragma unroll
        for (int t = 0; t < 4; t++) sc = fmaf(k4[j][t], q4[t], sc);
      }
    }
    if (bias_row) sc += bias_row[key];
    pl[key] = sc;
    mx = fmaxf(mx, sc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float m_use = (mx == -INFINITY) ? 0.f : mx;
  float sum = 0.f;
  for (int key = lane; key < a.L; key += 64) {
    const float p = __expf(pl[key] - m_use);
    pl[key] = p;
    sum += p;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const int Lp = (a.L + 3) & ~3;
  if (lane < Lp - a.L) pl[a.L + lane] = 0.f;   // zero pad to a multiple of 4 keys
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own LDS writes have landed (wave-private slice)
  float acc = 0.f;
  for (int key = 0; key < Lp; key += 4) {
    const f32x4 p4 = *reinterpret_cast<const f32x4 *>(pl + key);
    float vv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) vv[j] = a.v[at_row(a, seq, min(key + j, a.L - 1)) * a.ldv + hc + lane];
#pragma unroll
    for (int j = 0; j < 4; j++) acc = fmaf(p4[j], vv[j], acc);
  }
  const float res = acc / sum;
  if (a.out_planes) {
    unsigned p1, p2, p3;
    s3_split_pair(res, 0.f, p1, p2, p3);
    const size_t dst = s3_pack_off(q_row, hc + lane, a.plane_elems / (a.nheads * 64));
    a.out_planes[dst] = (unsigned short)p1;
    a.out_planes[a.plane_elems + dst] = (unsigned short)p2;
    a.out_planes[2 * a.plane_elems + dst] = (unsigned short)p3;
  } else {
    a.out[q_row * a.ldo + hc + lane] = res;
  }
}

// LDS: NBUF x UP regions of one (K tile | V tile); UP = units a workgroup can span (1 when nq % ipw == 0, else 2).
static_assert(4 * AT_RAG_SLICE <= AT_TILE, "ragged-row slices must fit the smallest LDS configuration");

template <int NBUF, int UP, bool S3>
__global__ __launch_bounds__(256, (NBUF * UP >= 4 || (S3 && UP > 1)) ? 2 : 3) void attn_f32_kernel(AttnArgs a) {
  constexpr int KF = S3 ? AT_K3_FLOATS : AT_KT * AT_LDK;   // floats of the K part of a staged tile
  constexpr int VF = S3 ? AT_V3_FLOATS : AT_KT * AT_LDV;   // ... and of its V part
  constexpr int TILE = KF + VF;
  __shared__ __attribute__((aligned(16))) float smem[NBUF * UP * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  int p = blockIdx.x;
  if (p < a.n_rag_wgs) {  // ---- ragged query rows on the VALU ----
    const int it = p * 4 + wave, n_units = a.nseq * a.nheads;
    if (it < n_units * a.rag) attn_ragged_row(a, it / a.rag, a.nq * 32 + it % a.rag, smem + wave * AT_RAG_SLICE, lane);
    return;
  }
  p -= a.n_rag_wgs;
  // XCD-aware order: physical block p runs on XCD p % 8; logical workgroups [g*group, (g+1)*group) share units and are
  // dealt to one XCD in consecutive slots
  const int wg = (((p >> 3) / a.group) * 8 + (p & 7)) * a.group + (p >> 3) % a.group;
  if (wg >= a.n_wgs) return;

  // Start-up stagger.  A launch is short (tens of microseconds) and every workgroup of the first dispatch round starts
  // at the same instant, so the 2-3 workgroups sharing a CU would run in lock step: all in their Q / first-tile loads
  // together, all in their MFMA phases together, all storing together -- the matrix pipe idles through both memory
  // phases.  Delaying the workgroup in hardware wave slot s of its SIMD by s * stagger_cycles puts one workgroup's
  // memory phases beside the others' MFMA phases.  The slot id is read from the hardware (HW_ID.wave_id), nothing is
  // assumed about placement; it only shifts time, never results.
  if (a.stagger_cycles > 0 && p < a.first_round) {
    const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 7u;   // HW_REG_HW_ID bits [3:0] = wave slot in the SIMD
    const unsigned s0 = __builtin_amdgcn_readfirstlane(slot);
    if (s0 > 0) {
      const long long t_end = (long long)__builtin_amdgcn_s_memtime() + (long long)s0 * a.stagger_cycles;
      while ((long long)__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(32);
    }
  }

  AT_STAMP(0);
  const int item0 = wg * a.ipw;
  const int uA = item0 / a.nqp;
  const int last = min(item0 + a.ipw, a.n_items) - 1;
  const bool two = UP > 1 && (last / a.nqp) != uA;           // workgroup-uniform: a second unit is present
  const int item = item0 + wave;
  const int unit_i = item / a.nqp, rb_i = item - unit_i * a.nqp;
  const bool active = wave < a.ipw && item < a.n_items && rb_i < a.nq;   // wave-uniform (rb_i >= nq: a padding slot of mode 7)
  const int unit = active ? unit_i : uA;
  const int rb = active ? rb_i : 0;
  const int mine = (UP > 1 && unit != uA) ? 1 : 0;           // which staged region this wave reads
  const int head = unit % a.nheads, seq = unit / a.nheads;

  const int lq = lane & 31, lh = lane >> 5;
  const int q_tok = rb * 32 + lq;  // this lane's query token
  const bool q_ok = active && q_tok < a.L;
  const int q_tok_c = q_tok < a.L ? q_tok : a.L - 1;
  const long q_row = at_row(a, seq, q_tok_c);
  const int hc = head * 64;

  // Q^T fragment.  fp32 MFMA: lane (q, h) holds Q[q][(2c+h)*4 + t], c=0..7, t=0..3  (pairs with the K read below).
  // S3: lane (q, h) holds, per 16-d step c = 0..3, the three bf16 pieces of Q[q][16c + 8h + 0..7].
  float qf[S3 ? 1 : 32];
  at_bf16x8 q3[S3 ? 4 : 1][3];
  {
    const float *qp = a.q + q_row * a.ldq + hc;
    if constexpr (S3) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(qp + 16 * c + 8 * lh), v1 = *reinterpret_cast<const f32x4 *>(qp + 16 * c + 8 * lh + 4);
        at_u32x4 p1, p2, p3;
        unsigned a0, b0, c0;
        // S3: scores live in the LOG2 domain -- q carries scale * log2(e), so the softmax below is exp2 (v_exp_f32) with no multiply
        const float qs = a.scale * AT_LOG2E;
        s3_split_pair(v0[0] * qs, v0[1] * qs, a0, b0, c0); p1[0] = a0, p2[0] = b0, p3[0] = c0;
        s3_split_pair(v0[2] * qs, v0[3] * qs, a0, b0, c0); p1[1] = a0, p2[1] = b0, p3[1] = c0;
        s3_split_pair(v1[0] * qs, v1[1] * qs, a0, b0, c0); p1[2] = a0, p2[2] = b0, p3[2] = c0;
        s3_split_pair(v1[2] * qs, v1[3] * qs, a0, b0, c0); p1[3] = a0, p2[3] = b0, p3[3] = c0;
        q3[c][0] = __builtin_bit_cast(at_bf16x8, p1);
        q3[c][1] = __builtin_bit_cast(at_bf16x8, p2);
        q3[c][2] = __builtin_bit_cast(at_bf16x8, p3);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const f32x4 v4 = *reinterpret_cast<const f32x4 *>(qp + (2 * c + lh) * 4);
#pragma unroll
        for (int t = 0; t < 4; t++) qf[c * 4 + t] = v4[t] * a.scale;
      }
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

  // K/V staging: 32 keys x 16 float4 per operand = 512 float4 each = 2 + 2 per thread and unit.  Loads are
  // unconditional: keys past L are clamped to the last token -- their scores are forced to -inf below, so P = 0 and the
  // (finite) clamped K/V values never contribute.
  const int kr = tid >> 4, c4 = tid & 15;     // rows kr and kr + 16, 16-byte chunk c4
  const int seqA = uA / a.nheads, hcA = (uA % a.nheads) * 64;
  const int uB = min(uA + 1, a.nseq * a.nheads - 1);
  const int seqB = uB / a.nheads, hcB = (uB % a.nheads) * 64;
  // V staging of the S3 variant: thread -> (d = tid & 63, key octet = wave): the eight keys of MFMA k-slots (h, 0..7) of group G,
  // (G, h) = (wave >> 1, wave & 1)
  struct VRegs {
    f32x4 v4[S3 ? 1 : 2];
    float v1[S3 ? 8 : 1];
  };
  f32x4 rk[2];
  VRegs rv;
  auto load_v = [&](VRegs &dst, int key0, int sq, int hcol) {
    if constexpr (S3) {
      const int kbase = key0 + 16 * (wave >> 1) + 4 * (wave & 1);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int tok = min(kbase + (j & 3) + 8 * (j >> 2), a.L - 1);   // wave-uniform
        dst.v1[j] = a.v[at_row(a, sq, tok) * a.ldv + hcol + (tid & 63)];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int tok = min(key0 + kr + 16 * i, a.L - 1);
        dst.v4[i] = *reinterpret_cast<const f32x4 *>(a.v + at_row(a, sq, tok) * a.ldv + hcol + c4 * 4);
      }
    }
  };
  auto issue_stage = [&](int key0, int sq, int hcol) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int tok = min(key0 + kr + 16 * i, a.L - 1);
      const long row = at_row(a, sq, tok);
      rk[i] = *reinterpret_cast<const f32x4 *>(a.k + row * a.ldk + hcol + c4 * 4);
    }
    load_v(rv, key0, sq, hcol);
  };
  // K/V registers of one staged tile -> an LDS region (S3: K goes in as three bf16 planes, 8 bytes per thread, row and plane)
  auto store_kv = [&](float *region, const f32x4 *k2, const VRegs &v2) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      if constexpr (S3) {
        unsigned a0, b0, c0, a1, b1, c1;
        s3_split_pair(k2[i][0], k2[i][1], a0, b0, c0);
        s3_split_pair(k2[i][2], k2[i][3], a1, b1, c1);
        unsigned char *kb = reinterpret_cast<unsigned char *>(region) + (kr + 16 * i) * AT_K3_ROWB + c4 * 8;
        *reinterpret_cast<at_u32x2 *>(kb) = at_u32x2{a0, a1};
        *reinterpret_cast<at_u32x2 *>(kb + AT_KT * AT_K3_ROWB) = at_u32x2{b0, b1};
        *reinterpret_cast<at_u32x2 *>(kb + 2 * AT_KT * AT_K3_ROWB) = at_u32x2{c0, c1};
      } else {
        *reinterpret_cast<f32x4 *>(region + (kr + 16 * i) * AT_LDK + c4 * 4) = k2[i];
        *reinterpret_cast<f32x4 *>(region + KF + (kr + 16 * i) * AT_LDV + c4 * 4) = v2.v4[i];
      }
    }
    if constexpr (S3) {   // V^T planes: row d = tid & 63, 16-byte slot = key octet (wave)
      at_u32x4 p1, p2, p3;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        unsigned x1, x2, x3;
        s3_split_pair(v2.v1[2 * j], v2.v1[2 * j + 1], x1, x2, x3);
        p1[j] = x1, p2[j] = x2, p3[j] = x3;
      }
      unsigned char *vb = reinterpret_cast<unsigned char *>(region + KF) + (tid & 63) * AT_V3_ROWB + wave * 16;
      *reinterpret_cast<at_u32x4 *>(vb) = p1;
      *reinterpret_cast<at_u32x4 *>(vb + 64 * AT_V3_ROWB) = p2;
      *reinterpret_cast<at_u32x4 *>(vb + 2 * 64 * AT_V3_ROWB) = p3;
    }
  };
  auto store_stage = [&](float *region) { store_kv(region, rk, rv); };
  auto region = [&](int buf, int u) { return smem + (buf * UP + u) * TILE; };

  // Ragged key tail: L = 289 and 545 are 9 / 17 full key tiles plus ONE key.  A 32-key MFMA tile for one key is
  // 97 % padding, so a tail of <= 2 keys is folded into the online softmax on the VALU instead (a rank-1 update).
  const int rem = a.L % AT_KT;
  const bool tail_valu = rem > 0 && rem <= 2 && a.L > AT_KT;
  const int ntiles = tail_valu ? a.L / AT_KT : (a.L + AT_KT - 1) / AT_KT;

  // one tile of this wave's item: S^T, online softmax, O^T update, reading region `reg`
  auto compute_s = [&](const float *sK, f32x16 &s) {
    if constexpr (S3) {
      // four 16-d steps, six bf16 MFMAs each (terms smallest first, as gemm_split.hip), ONE accumulation chain: a dependent
      // v_mfma_f32_32x32x16_bf16 issues back to back, so a second accumulator would only cost its zeroing and the final add
#pragma unroll
      for (int e = 0; e < 16; e++) s[e] = 0.f;
      const unsigned char *kb = reinterpret_cast<const unsigned char *>(sK) + lq * AT_K3_ROWB + lh * 16;
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const at_bf16x8 k1 = *reinterpret_cast<const at_bf16x8 *>(kb + c * 32);
        const at_bf16x8 k2 = *reinterpret_cast<const at_bf16x8 *>(kb + AT_KT * AT_K3_ROWB + c * 32);
        const at_bf16x8 k3 = *reinterpret_cast<const at_bf16x8 *>(kb + 2 * AT_KT * AT_K3_ROWB + c * 32);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k3, q3[c][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q3[c][2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, q3[c][1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2, q3[c][0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q3[c][1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, q3[c][0], s, 0, 0, 0);
      }
      return;
    } else {
    f32x16 sb;
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = 0.f, sb[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; c++) {   // two independent accumulation chains (dims 0-31 | 32-63), interleaved
        const f32x4 ka = *reinterpret_cast<const f32x4 *>(&sK[lq * AT_LDK + (2 * c + lh) * 4]);
        const f32x4 kb = *reinterpret_cast<const f32x4 *>(&sK[lq * AT_LDK + (2 * (c + 4) + lh) * 4]);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], qf[c * 4 + t], s, 0, 0, 0);
          sb = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[t], qf[(c + 4) * 4 + t], sb, 0, 0, 0);
        }
      }
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] += sb[e];
    }
  };
  auto softmax_pv = [&](const float *sV, f32x16 &s, int key0) {
    // lane holds keys key0 + (e&3) + 8*(e>>2) + 4*lh  for its query
    if (bias_base) {
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int kb = min(key0 + 8 * g4 + 4 * lh, a.L - 4);  // L % 4 == 0 with a bias; clamped reads are masked below
        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bias_base + kb);
#pragma unroll
        for (int t = 0; t < 4; t++) s[g4 * 4 + t] = S3 ? fmaf(b4[t], AT_LOG2E, s[g4 * 4 + t]) : s[g4 * 4 + t] + b4[t];
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
    if constexpr (S3) {
      // log2-domain online softmax with a DEFERRED running maximum (AT_DEFER): most tiles skip the rescale of O and l entirely
      float psum = 0.f;
      if (__any(mx > m_run + AT_DEFER || (m_run == -INFINITY && mx != -INFINITY))) {   // wave-uniform
        const float m_new = fmaxf(m_run, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // whole row masked so far
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf -> 0
        l_run *= alpha;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 16; e++) o0[e] *= alpha, o1[e] *= alpha;
      }
      const float m_use = (m_run == -INFINITY) ? 0.f : m_run;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        s[e] = __builtin_amdgcn_exp2f(s[e] - m_use);
        psum += s[e];
      }
      l_run += psum;
    } else {
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
    }
    // ---- O^T[d][q] += sum_key V[key][d] * P[q][key];  k-step e pairs key (e&3)+8*(e>>2)+4*lh ----
    if constexpr (S3) {
      // P (this lane's 16 values, in [0, 1]) -> three bf16 pieces; elements 8 G .. 8 G + 7 are k-slots (lh, 0..7) of group G
      at_u32x4 pp[2][3];
#pragma unroll
      for (int G = 0; G < 2; G++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          unsigned x1, x2, x3;
          s3_split_pair(s[8 * G + 2 * j], s[8 * G + 2 * j + 1], x1, x2, x3);
          pp[G][0][j] = x1, pp[G][1][j] = x2, pp[G][2][j] = x3;
        }
      const unsigned char *vb = reinterpret_cast<const unsigned char *>(sV) + lq * AT_V3_ROWB + lh * 16;
#pragma unroll
      for (int G = 0; G < 2; G++) {
        const at_bf16x8 p1 = __builtin_bit_cast(at_bf16x8, pp[G][0]), p2 = __builtin_bit_cast(at_bf16x8, pp[G][1]),
                        p3 = __builtin_bit_cast(at_bf16x8, pp[G][2]);
#pragma unroll
        for (int o = 0; o < 2; o++) {
          const unsigned char *vo = vb + o * 32 * AT_V3_ROWB + G * 32;
          const at_bf16x8 v1 = *reinterpret_cast<const at_bf16x8 *>(vo);
          const at_bf16x8 v2 = *reinterpret_cast<const at_bf16x8 *>(vo + 64 * AT_V3_ROWB);
          const at_bf16x8 v3 = *reinterpret_cast<const at_bf16x8 *>(vo + 2 * 64 * AT_V3_ROWB);
          f32x16 &t = o ? o1 : o0;
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v3, p1, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p3, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, p2, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v2, p1, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p2, t, 0, 0, 0);
          t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v1, p1, t, 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const float v0 = sV[key * AT_LDV + lq], v1 = sV[key * AT_LDV + 32 + lq];
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, s[e], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, s[e], o1, 0, 0, 0);
      }
    }
  };

  if constexpr (NBUF == 2) {
    // prologue: tile 0 of both units into buffer 0, then tile 1 of unit A into the registers
    issue_stage(0, seqA, hcA);
    store_stage(region(0, 0));
    if (two) {
      issue_stage(0, seqB, hcB);
      store_stage(region(0, UP - 1));
    }
    if (ntiles > 1) issue_stage(AT_KT, seqA, hcA);
    __syncthreads();
    AT_STAMP(1);
    for (int kt = 0; kt < ntiles; ++kt) {
      const int cur = kt & 1, nxt = cur ^ 1, key0 = kt * AT_KT;
      const float *reg = region(cur, mine);
      f32x16 s;
      if (active) compute_s(reg, s);
      if (kt + 1 < ntiles) {   // unit A's tile kt+1 (loaded one tile ago) -> the other buffer; then the registers go to B / A's tile kt+2
        store_stage(region(nxt, 0));
        if (two) issue_stage(key0 + AT_KT, seqB, hcB);
        else if (kt + 2 < ntiles) issue_stage(key0 + 2 * AT_KT, seqA, hcA);
      }
      if (active) softmax_pv(reg + KF, s, key0);
      if (two && kt + 1 < ntiles) {
        store_stage(region(nxt, UP - 1));
        if (kt + 2 < ntiles) issue_stage(key0 + 2 * AT_KT, seqA, hcA);
      }
      // one barrier per tile: LDS traffic only (the staged global loads stay in flight across it)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    f32x4 rk2[2];           // second unit's tile: both must be written between the two barriers
    VRegs rv2;
    auto issue_stage2 = [&](int key0) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int tok = min(key0 + kr + 16 * i, a.L - 1);
        const long row = at_row(a, seqB, tok);
        rk2[i] = *reinterpret_cast<const f32x4 *>(a.k + row * a.ldk + hcB + c4 * 4);
      }
      load_v(rv2, key0, seqB, hcB);
    };
    issue_stage(0, seqA, hcA);
    if (two) issue_stage2(0);
    AT_STAMP(1);
    for (int kt = 0; kt < ntiles; ++kt) {
      const int key0 = kt * AT_KT;
      store_stage(region(0, 0));
      if (two) store_kv(region(0, UP - 1), rk2, rv2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + 1 < ntiles) {  // next tile's loads fly during this tile's 64 MFMAs
        issue_stage(key0 + AT_KT, seqA, hcA);
        if (two) issue_stage2(key0 + AT_KT);
      }
      if (active) {
        const float *reg = region(0, mine);
        f32x16 s;
        compute_s(reg, s);
        softmax_pv(reg + KF, s, key0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // tile fully consumed before the next store
    }
  }
  AT_STAMP(2);
  if (!active) return;

  if (tail_valu) {
    for (int key = ntiles * AT_KT; key < a.L; ++key) {
      const long row = at_row(a, seq, key);
      const float *kp = a.k + row * a.ldk + hc, *vp = a.v + row * a.ldv + hc;
      float sc = 0.f;  // this lane's half of q . k (dims (2c+lh)*4 + t), completed by the cross-half shuffle
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const f32x4 k4 = *reinterpret_cast<const f32x4 *>(kp + (2 * c + lh) * 4);
        if constexpr (S3) {   // the fp32 q is not kept in registers: one more (L2-hot) read for the single tail key
          const f32x4 q4 = *reinterpret_cast<const f32x4 *>(a.q + q_row * a.ldq + hc + (2 * c + lh) * 4);
#pragma unroll
          for (int t = 0; t < 4; t++) sc = fmaf(k4[t], q4[t] * (a.scale * AT_LOG2E), sc);   // log2 domain, as the MFMA tiles
        } else {
#pragma unroll
          for (int t = 0; t < 4; t++) sc = fmaf(k4[t], qf[c * 4 + t], sc);
        }
      }
      sc += __shfl_xor(sc, 32);
      if (bias_base) sc += S3 ? bias_base[key] * AT_LOG2E : bias_base[key];
      const float m_new = fmaxf(m_run, sc);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = S3 ? __builtin_amdgcn_exp2f(m_run - m_use) : __expf(m_run - m_use);
      const float pk = S3 ? __builtin_amdgcn_exp2f(sc - m_use) : __expf(sc - m_use);
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
      if (a.out_planes) {
        const size_t prows = a.plane_elems / (a.nheads * 64);   // slice-major planes (split3.h): columns hc + d and hc + 32 + d are one slice apart
        const size_t dst = s3_pack_off(q_row, hc + d, prows);
        s3_store4(a.out_planes, a.plane_elems, dst, w0);
        s3_store4(a.out_planes, a.plane_elems, dst + prows * 32, w1);
      } else {
        *reinterpret_cast<f32x4 *>(op + d) = w0;
        *reinterpret_cast<f32x4 *>(op + 32 + d) = w1;
      }
    }
  }
  AT_STAMP(3);
}

template <typename K>
static inline void launch_attn(K kernel, unsigned grid, hipStream_t st, const AttnArgs &a) {
  kernel<<<grid, 256, 0, st>>>(a);
}

static int attention_any(const float *d_q, int ldq, const float *d_k, int ldk, const float *d_v, int ldv, float *d_out, int ldo,
                         uint16_t *d_planes, long plane_rows, int L, int nseq, int nheads, const int32_t *d_rowmap,
                         const float *d_bias, const int32_t *d_biasvar, float scale, const sgic_launch_opts *opts,
                         sgic_stream_t stream) {
  SGIC_REQUIRE(d_q && d_k && d_v && (d_out || d_planes) && L > 0 && nseq > 0 && nheads > 0, "args");
  SGIC_REQUIRE((ldq & 3) == 0 && (ldk & 3) == 0 && (ldv & 3) == 0 && (ldo & 3) == 0, "row strides must be multiples of 4");
  SGIC_REQUIRE(ldq >= nheads * 64 && ldk >= nheads * 64 && ldv >= nheads * 64 && (d_planes || ldo >= nheads * 64), "head_dim is 64");
  SGIC_REQUIRE((((uintptr_t)d_q | (uintptr_t)d_k | (uintptr_t)d_v | (uintptr_t)d_out) & 15) == 0 && ((uintptr_t)d_planes & 7) == 0, "16-byte alignment");
  SGIC_REQUIRE(!d_planes || plane_rows >= (long)nseq * L, "plane rows");
  SGIC_REQUIRE(!d_bias || (L & 3) == 0, "bias needs L % 4 == 0");
  const int mode_all = opts ? opts->attn_mode : 0;
  // attn_mode: low 3 bits: 0 = default (see below); odd = single LDS buffer (two barriers per tile), even = double-buffered LDS
  // (one barrier per tile); modes 3,4 / 5,6 add the start-up stagger of 4096 / 8192 cycles per hardware wave slot.
  // bit 3 (+8): S^T = K . Q^T as a bf16x3 split product on the bf16 matrix pipe (fp32-accurate; results differ in the last bits
  // from the fp32-MFMA variant, and are again identical for every mode of the same variant)
  // mode 7: THREE row blocks per workgroup with the unit's row blocks padded to a multiple of three (single buffer, no stagger): a
  // workgroup never spans two units, so it runs the one-unit kernel at three workgroups per CU -- 9 computing waves per CU.  For
  // L = 289 (9 row blocks x 512 units at batch 32) that is exactly two rounds of the 768 resident workgroups instead of 2.25 rounds
  // of 512 (whose last round runs a quarter full); L = 545 (17 -> 18 slots): three rounds instead of 3.19.
  SGIC_REQUIRE(mode_all >= 0 && mode_all <= 15, "attn_mode 0..7 (+8)");
  const bool s3 = (mode_all & 8) != 0;
  const int mode = mode_all & 7;
  const long units = (long)nseq * nheads;
  const int rem = L % 32;
  const int rag = (rem >= 1 && rem <= 2 && L >= 64 && L <= AT_RAG_MAXL) ? rem : 0;   // ragged query rows -> VALU path
  const int nq = rag ? L / 32 : (L + 31) / 32;
  const bool three = mode == 7 && nq >= 2;
  const int ipw = three ? 3 : (nq == 1 ? 2 : 4);
  const int nqp = three ? (nq + 2) / 3 * 3 : nq;
  const long n_items = units * nqp;
  SGIC_REQUIRE(n_items < (1l << 30), "too many (sequence, head, row block) items");
  const int n_wgs = (int)((n_items + ipw - 1) / ipw);
  const int group = (4 * nqp) / ipw;                     // workgroups that cover 4 whole units
  const long groups = (n_wgs + group - 1) / group;
  const long grid_mfma = ((groups + 7) / 8) * 8 * group;
  const int n_rag_wgs = (int)((units * rag + 3) / 4);
  const bool up2 = (nqp % ipw) != 0;
  AttnArgs a{d_q, d_k, d_v, d_out, ldq, ldk, ldv, ldo, L, nseq, nheads, d_rowmap, d_bias, d_biasvar, scale,
             nq, ipw, nqp, (int)n_items, n_wgs, group, rag, n_rag_wgs, 0, 0, d_planes, plane_rows * nheads * 64};
  const unsigned grid = (unsigned)(grid_mfma + n_rag_wgs);
  hipStream_t st = to_stream(stream);
  // default (mode 0), from the round-2 measurements (tools/bench_attn.py, tools/micro/attn_stamps.hip): a single K/V buffer
  // (3 workgroups per CU) with the 8192-cycle start-up stagger when workgroups span two units (L = 289, 545), the plain
  // single buffer otherwise (L = 256 windows, the CLIP towers)
  const int eff = three ? 1 : (mode ? (mode == 7 ? 1 : mode) : (up2 ? 5 : 1));
  const bool single = (eff & 1) != 0;
  // first dispatch round = what is resident at once: workgroups per CU (LDS / register limited) x 256 CUs
  a.stagger_cycles = eff > 2 ? 4096 * ((eff - 1) / 2) : 0;
  const int per_cu = (!single && up2) ? 2 : 3;
  a.first_round = per_cu * 256;
  if (s3) {
    if (up2) {
      if (single) launch_attn(attn_f32_kernel<1, 2, true>, grid, st, a);
      else launch_attn(attn_f32_kernel<2, 2, true>, grid, st, a);
    } else {
      if (single) launch_attn(attn_f32_kernel<1, 1, true>, grid, st, a);
      else launch_attn(attn_f32_kernel<2, 1, true>, grid, st, a);
    }
  } else if (up2) {
    if (single) launch_attn(attn_f32_kernel<1, 2, false>, grid, st, a);
    else launch_attn(attn_f32_kernel<2, 2, false>, grid, st, a);
  } else {
    if (single) launch_attn(attn_f32_kernel<1, 1, false>, grid, st, a);
    else launch_attn(attn_f32_kernel<2, 1, false>, grid, st, a);
  }
  return sgic::check_launch("attn_f32_kernel");
}

extern "C" int sgic_attention_f32(const float *d_q, int ldq, const float *d_k, int ldk, const float *d_v, int ldv,
                                  float *d_out, int ldo, int L, int nseq, int nheads, const int32_t *d_rowmap,
                                  const float *d_bias, const int32_t *d_biasvar, float scale, const sgic_launch_opts *opts,
                                  sgic_stream_t stream) {
  SGIC_REQUIRE(d_out, "out");
  return attention_any(d_q, ldq, d_k, ldk, d_v, ldv, d_out, ldo, nullptr, 0, L, nseq, nheads, d_rowmap, d_bias, d_biasvar, scale,
                       opts, stream);
}

// the same attention with its output written directly as bf16x3 planes [3][rows][nheads*64] (rows >= nseq*L: the row space
// of the row map), i.e. as the A operand of the out-projection when that runs as a split GEMM
extern "C" int sgic_attention_split3_f32(const float *d_q, int ldq, const float *d_k, int ldk, const float *d_v, int ldv,
                                         uint16_t *d_out_planes, long rows, int L, int nseq, int nheads,
                                         const int32_t *d_rowmap, const float *d_bias, const int32_t *d_biasvar, float scale,
                                         const sgic_launch_opts *opts, sgic_stream_t stream) {
  SGIC_REQUIRE(d_out_planes, "planes");
  return attention_any(d_q, ldq, d_k, ldk, d_v, ldv, nullptr, 0, d_out_planes, rows, L, nseq, nheads, d_rowmap, d_bias, d_biasvar,
                       scale, opts, stream);
}
