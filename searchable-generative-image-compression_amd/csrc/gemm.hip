// fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// This is THE dominant kernel of the codec (>= 95 % of the FLOPs: every nn.Linear, 1x1 conv, the patch-embed /
// 2x2 convs as im2col GEMMs, the VQGAN 3x3 convs as implicit GEMMs; reference call sites titok/blocks.py:37-64,
// blocks/swin_transformer.py:94-156, models/cross_blocks.py:75-98, blocks/conv_blocks.py:71-81,
// blocks/dcvc.py:28-54, taming/modules/diffusionmodules/model.py:38-192).  The reference computes in fp32, so
// we use the exact-fp32 MFMA v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, 157.3 TFLOP/s chip peak); its result is
// bit-for-bit a k-ordered fmaf chain, and the k order here is fixed by K alone (no split-K / stream-K), so every
// output row is independent of M, of the tile shape and of the batch it sits in (batch-invariant).
//
// Tiling (wave64): 256 threads = 4 waves as 2(M) x 2(N); wave tile (32*WM) x (32*WN) MFMA blocks, workgroup tile
// (64*WM) x (64*WN): 128x128 (2,2), 128x64 (2,1), 64x64 (1,1); BK = 32.
//  * staging: A/W tiles global -> VGPR (float4, UNCONDITIONAL loads; out-of-range rows are clamped -- they only
//    feed accumulator rows/columns the epilogue never stores) -> LDS with a 36-float row stride (the 16 lanes of a
//    ds_read_b128 group hit 16 distinct 16-byte slots: conflict-free);
//  * pipeline (NBUF = 2): double-buffered LDS, ONE barrier per K-step; tile k+2's loads are issued at the top of
//    step k and first touched by the ds_write one step later; MFMA fragments are double-buffered in registers and
//    the barrier sits before the last sub-step, after which the next tile's first fragments are prefetched.
//    NBUF = 1: one LDS buffer, two barriers per step, but 3-4 workgroups per CU;
//  * operands: ds_read_b128 (4 consecutive k per lane); MFMA step t uses element t of both fragments, i.e.
//    lanes 0-31 feed k = 8s+t and lanes 32-63 k = 8s+4+t (a fixed permutation of the k order);
//  * epilogue: per-wave LDS transpose, whole 128/256-byte row segments stored as dwordx4 with bias ->
//    activation -> residual fused;
//  * order: XCD-aware bijective block remap + grouped (8 m-tiles x all n-tiles) walk for L2 reuse;
//  * MIXED launch: the bulk of the rows uses 128x128 tiles in whole rounds of the 256 CUs, the remaining rows use
//    64x64 tiles inside the SAME launch, so the last round of workgroups is short instead of leaving most CUs idle
//    (M = 9248 = 72.25 tiles is never a multiple of the machine).
#include <math.h>

#include <vector>

#include <hip/hip_ext.h>

#include "common.h"
#include "gemm_act.h"
#include "profiler.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BK 32
#define LDS_LD 36

// Diagnostic build only (tools/micro/gemm_stamps.hip defines GEMM_STAMPS): per workgroup, s_memrealtime at start / end and
// the main-loop and epilogue cycles of thread 0, into a buffer nothing else reads.  The product build has no stamps.
#ifdef GEMM_STAMPS
__device__ long long gemm_stamps[8 * 4096];   // per block: [0] start(real) [1] end(real) [2] loop cycles [3] epilogue cycles [4] tiles [5] life cycles
#define GS_NOW() ((long long)__builtin_amdgcn_s_memtime())
#endif

struct GemmArgs {
  const float *A;
  const float *W;
  const float *bias;  // [N] or null
  const float *R;     // residual [M, ldr] or null (added after the activation)
  float *C;
  int M, N, K;
  int lda, ldw, ldr, ldc;
  int act;
  // optional row maps  row(m) = (m / seg) * seg_stride + (m % seg)  (seg == 0: identity) so a GEMM can
  // read / write the [:, a:b] token slice of an (n, L, C) buffer in place (models/cross_blocks.py:87-94)
  int a_seg, a_seg_stride, c_seg, c_seg_stride;
  int vec_epilogue;  // C/R/bias rows are float4-addressable (N, ldc, ldr % 4 == 0, 16-byte aligned bases)
  // batched GEMM: blockIdx.y = batch index, element strides (0 = shared operand)
  long sA, sW, sC, sR;
  // implicit-GEMM 3x3 convolution (stride 1, pad 1): A is a zero-halo NHWC buffer [B, H+2, W+2, C], logical row
  // m = (b,y,x) of the output, K = 9*C ordered (ky,kx,c); conv_C == 0 disables.  C % 32 == 0 so a 32-wide K
  // slice never straddles a tap (taming ResnetBlock / Upsample / conv_in / conv_out, model.py:38-137,436-537)
  int conv_C, conv_H, conv_W;
  int stagger_cycles, per_cu;
  // tile regions: rows [0, m_split) are covered by `big_blocks` workgroups of the kernel's primary tile, rows
  // [m_split, M) by 64x64 tiles (MIXED kernels only; otherwise m_split == M)
  int m_split, big_blocks;
};

// Output tiles of one workgroup.  bid = workgroup index inside its region, rows [row_base, row_end) x all N.
// PERSIST: the workgroup walks tiles bid, bid + step, ... ; the next tile's first K slice is requested BEFORE the
// current tile's epilogue, so the store phase, the workgroup hand-over and the first-load latency of a
// one-tile-per-workgroup launch disappear from the MFMA pipe's critical path.
template <int WM, int WN, bool KTAIL, int NBUF, bool PERSIST>
__device__ __forceinline__ void gemm_tile(const GemmArgs &g, int bid, int step, int row_base, int row_end, float *smem) {
  constexpr int TBM = 64 * WM, TBN = 64 * WN;       // workgroup tile
  constexpr int NA4 = TBM / 32, NB4 = TBN / 32;     // float4 staged per thread (rows x 8 chunks / 256 threads)
  float(*sA)[TBM * LDS_LD] = reinterpret_cast<float(*)[TBM * LDS_LD]>(smem);
  float(*sB)[TBN * LDS_LD] = reinterpret_cast<float(*)[TBN * LDS_LD]>(smem + NBUF * TBM * LDS_LD);

  // ---- tile order: XCD-aware bijective remap (consecutive workgroup ids are dealt round-robin over the 8 XCDs,
  // ids congruent mod 8 share an L2), then a grouped walk (8 m-tiles x all n-tiles, m fastest) so the ~64
  // workgroups resident on one XCD cover an ~8x8 patch of tiles and reuse each other's A / W k-slices in L2.
  const int tiles_m = (row_end - row_base + TBM - 1) / TBM, tiles_n = (g.N + TBN - 1) / TBN;
  const int nwg = tiles_m * tiles_n;
  int m0, n0;
  auto locate = [&](int t) {
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, within = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * tiles_n;
    const int group = t / per_group, first_m = group * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    const int in_group = t - group * per_group;
    m0 = row_base + (first_m + in_group % gsz) * TBM;
    n0 = (in_group / gsz) * TBN;
  };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * (32 * WM), wn = (wave & 1) * (32 * WN);
  const int srow = tid >> 3, schunk = tid & 7;  // staging map: rows srow + 32*i, 16-byte chunk schunk of the K slice
  f32x4 ra[NA4], rb[NB4];
  const float *aptr[NA4];
  const float *wptr[NB4];
  auto setup = [&]() {
#pragma unroll
    for (int i = 0; i < NA4; i++) {
      const int am = min(m0 + srow + 32 * i, g.M - 1);
      size_t arow;
      if (g.conv_C) {  // top-left pixel of the 3x3 patch in the halo buffer
        const int hw = g.conv_H * g.conv_W, b = am / hw, r = am - b * hw, y = r / g.conv_W, x = r - y * g.conv_W;
        arow = ((size_t)b * (g.conv_H + 2) + y) * (g.conv_W + 2) + x;
      } else {
        arow = g.a_seg ? (size_t)(am / g.a_seg) * g.a_seg_stride + (am % g.a_seg) : (size_t)am;
      }
      aptr[i] = g.A + arow * g.lda;
    }
#pragma unroll
    for (int i = 0; i < NB4; i++) {
      const int wr = min(n0 + srow + 32 * i, g.N - 1);
      wptr[i] = g.W + (size_t)wr * g.ldw;
    }
  };
  if (PERSIST && bid >= nwg) return;
  locate(bid);
  setup();

  // KTAIL = false (K % 32 == 0, every large GEMM of the model): plain loads whose results are first touched by
  // the ds_write one K-step later, so their latency hides under a full MFMA phase.  KTAIL = true: the tail chunk
  // is clamped in address and zeroed by a select (this consumes the load early -- slow path, odd K only).
  auto issue_loads = [&](int k0) {
    if constexpr (!KTAIL) {
      int ka = k0;
      if (g.conv_C) {  // wave-uniform: which tap this K slice belongs to
        const int tap = k0 / g.conv_C, c0 = k0 - tap * g.conv_C, ky = tap / 3, kx = tap - 3 * ky;
        ka = (ky * (g.conv_W + 2) + kx) * g.conv_C + c0;
      }
#pragma unroll
      for (int i = 0; i < NA4; i++) ra[i] = *reinterpret_cast<const f32x4 *>(aptr[i] + ka + schunk * 4);
#pragma unroll
      for (int i = 0; i < NB4; i++) rb[i] = *reinterpret_cast<const f32x4 *>(wptr[i] + k0 + schunk * 4);
    } else {
      const int kk = k0 + schunk * 4;
      const bool kok = kk < g.K;  // K % 4 == 0, so a chunk is all-in or all-out
      const int kc = kok ? kk : g.K - 4;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NA4; i++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(aptr[i] + kc);
        ra[i] = kok ? v : z;
      }
#pragma unroll
      for (int i = 0; i < NB4; i++) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(wptr[i] + kc);
        rb[i] = kok ? v : z;
      }
    }
  };
  auto store_lds = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA4; i++) *reinterpret_cast<f32x4 *>(&sA[buf][(srow + 32 * i) * LDS_LD + schunk * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < NB4; i++) *reinterpret_cast<f32x4 *>(&sB[buf][(srow + 32 * i) * LDS_LD + schunk * 4]) = rb[i];
  };

  f32x16 acc[WM][WN];
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int aoff = (wm + lrow) * LDS_LD + lhalf * 4;
  const int boff = (wn + lrow) * LDS_LD + lhalf * 4;
  struct Frag {
    f32x4 a[WM], b[WN];
  };
  auto read_frag = [&](Frag &f, int buf, int s) {
#pragma unroll
    for (int i = 0; i < WM; i++) f.a[i] = *reinterpret_cast<const f32x4 *>(&sA[buf][aoff + i * 32 * LDS_LD + s * 8]);
#pragma unroll
    for (int j = 0; j < WN; j++) f.b[j] = *reinterpret_cast<const f32x4 *>(&sB[buf][boff + j * 32 * LDS_LD + s * 8]);
  };
  auto mfma_frag = [&](const Frag &f) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][t], f.b[j][t], acc[i][j], 0, 0, 0);
  };

  const int nk = (g.K + BK - 1) / BK;
  // Stagger (optional): the workgroups that land as the 2nd (3rd, 4th) resident of a CU in the first dispatch
  // round start a fraction of a tile later, so co-resident workgroups are not in their prologue / epilogue at the
  // same time.  Placement is only assumed for speed: ids congruent mod 8 share an XCD, 32 CUs per XCD fill in order.
  if (g.stagger_cycles > 0) {
    const int slot = ((int)blockIdx.x >> 3) / 32;
    if (slot > 0 && slot < g.per_cu) {
      const long long t_end = (long long)__builtin_amdgcn_s_memtime() + (long long)slot * g.stagger_cycles;
      while ((long long)__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(32);
    }
  }
  Frag f0, f1;
#ifdef GEMM_STAMPS
  long long gs_t0 = 0, gs_loop = 0, gs_epi = 0;
  int gs_tiles = 0;
  if (threadIdx.x == 0 && blockIdx.x < 4096) gemm_stamps[blockIdx.x * 8 + 0] = (long long)__builtin_amdgcn_s_memrealtime();
  const long long gs_mark = GS_NOW();
#endif
  issue_loads(0);
  for (;;) {
#ifdef GEMM_STAMPS
    gs_t0 = GS_NOW();
#endif
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
      for (int j = 0; j < WN; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
    if constexpr (NBUF == 2) {
      store_lds(0);
      if (nk > 1) issue_loads(BK);
      __syncthreads();
      read_frag(f0, 0, 0);
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1, nxt = cur ^ 1;
        if (kt + 1 < nk) {
          store_lds(nxt);  // tile kt+1; buffer nxt was last read before the previous step's barrier
          if (kt + 2 < nk) issue_loads((kt + 2) * BK);
        }
        read_frag(f1, cur, 1);
        mfma_frag(f0);
        read_frag(f0, cur, 2);
        mfma_frag(f1);
        read_frag(f1, cur, 3);
        mfma_frag(f0);
        // one barrier per K-step: LDS traffic only (the staged global loads stay in flight across it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) read_frag(f0, nxt, 0);
        mfma_frag(f1);
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) {
        store_lds(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) issue_loads((kt + 1) * BK);  // in flight during the MFMA phase below
        read_frag(f0, 0, 0);
        read_frag(f1, 0, 1);
        mfma_frag(f0);
        read_frag(f0, 0, 2);
        mfma_frag(f1);
        read_frag(f1, 0, 3);
        mfma_frag(f0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone done reading before the next store
        mfma_frag(f1);
      }
      __syncthreads();  // epilogue reuses the staging LDS
    }

#ifdef GEMM_STAMPS
    {
      const long long t_ = GS_NOW();
      gs_loop += t_ - gs_t0;
      gs_t0 = t_;
      gs_tiles++;
    }
#endif
    // the tile being finished, and (PERSIST) the next one: its first K slice is in flight during the epilogue
    const int em0 = m0, en0 = n0;
    bool more = false;
    if constexpr (PERSIST) {
      bid += step;
      more = bid < nwg;
      if (more) {
        locate(bid);
        setup();
        issue_loads(0);
      }
    }

    // ---- epilogue.  Accumulator layout: D[row = (e&3) + 8*(e>>2) + 4*lhalf][col = lrow] ----
    if (g.vec_epilogue) {
      // Wide stores: each wave transposes its accumulators 32 rows at a time through its private LDS slice (nobody
      // reads the staging buffers after the last barrier) and writes whole 128/256-byte row segments as dwordx4 --
      // 4x fewer store instructions than one dword per lane, which is what bounds a lock-stepped epilogue.
      // The activation / residual / row-map choices are hoisted out of the store loop (one straight-line variant
      // each): as per-element uniform branches they cost more than the stores themselves.
      constexpr int TW = 32 * WN;            // wave tile width (floats)
      constexpr int LPR = TW / 4;            // lanes per row (float4 each)
      constexpr int RPI = 64 / LPR;          // rows per store instruction
      float *ep = smem + wave * (32 * TW);
      const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
      const int n = en0 + wn + c4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (g.bias && n < g.N) bv = *reinterpret_cast<const f32x4 *>(g.bias + n);
      auto run = [&](auto act_c, auto res_c, auto map_c) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool HASR = decltype(res_c)::value != 0, MAPC = decltype(map_c)::value != 0;
        constexpr int NIT = 32 / RPI;  // store instructions per 32-row slab
        const bool colok = n < g.N;     // N % 4 == 0 in this path, so a float4 is all-in or all-out
        const int nc = colok ? n : 0;
#pragma unroll
        for (int i = 0; i < WM; i++) {
          // residual rows first (clamped, unconditional: their latency hides under the transpose), then the
          // transpose through LDS, then all row reads, and only the stores are predicated
          f32x4 rv[NIT];
          if constexpr (HASR) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
              const int m = min(em0 + wm + i * 32 + it * RPI + r0, row_end - 1);
              rv[it] = *reinterpret_cast<const f32x4 *>(g.R + (size_t)m * g.ldr + nc);
            }
          }
#pragma unroll
          for (int j = 0; j < WN; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) ep[((e & 3) + 8 * (e >> 2) + 4 * lhalf) * TW + j * 32 + lrow] = acc[i][j][e];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          constexpr int EB = NIT < 4 ? NIT : 4;  // rows in flight between the LDS read and the store
#pragma unroll
          for (int b = 0; b < NIT; b += EB) {
            f32x4 cv[EB];
#pragma unroll
            for (int q = 0; q < EB; ++q) cv[q] = *reinterpret_cast<const f32x4 *>(ep + ((b + q) * RPI + r0) * TW + c4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < EB; ++q) {
              const int it = b + q;
              const int m = em0 + wm + i * 32 + it * RPI + r0;
              f32x4 v = cv[q];
#pragma unroll
              for (int t = 0; t < 4; t++) v[t] = apply_act_c<ACT>(v[t] + bv[t]);
              if constexpr (HASR) v += rv[it];
              size_t crow = (size_t)m;
              if constexpr (MAPC) crow = (size_t)(m / g.c_seg) * g.c_seg_stride + (m % g.c_seg);
              if (colok && m < row_end) *reinterpret_cast<f32x4 *>(g.C + crow * g.ldc + n) = v;
            }
          }
        }
      };
      auto run_act = [&](auto act_c) {
        if (g.c_seg) {
          if (g.R) run(act_c, IntC<1>{}, IntC<1>{});
          else run(act_c, IntC<0>{}, IntC<1>{});
        } else {
          if (g.R) run(act_c, IntC<1>{}, IntC<0>{});
          else run(act_c, IntC<0>{}, IntC<0>{});
        }
      };
      switch (g.act) {
        case ACT_GELU: run_act(IntC<ACT_GELU>{}); break;
        case ACT_SILU: run_act(IntC<ACT_SILU>{}); break;
        case ACT_TANH: run_act(IntC<ACT_TANH>{}); break;
        case ACT_LRELU: run_act(IntC<ACT_LRELU>{}); break;
        default: run_act(IntC<ACT_NONE>{}); break;
      }
    } else {
#pragma unroll
      for (int j = 0; j < WN; j++) {
        const int n = en0 + wn + j * 32 + lrow;
        if (n >= g.N) continue;
        const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; i++) {
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int m = em0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
            if (m < row_end) {
              float v = apply_act(acc[i][j][e] + bv, g.act);
              if (g.R) v += g.R[(size_t)m * g.ldr + n];
              const size_t crow = g.c_seg ? (size_t)(m / g.c_seg) * g.c_seg_stride + (m % g.c_seg) : (size_t)m;
              g.C[crow * g.ldc + n] = v;
            }
          }
        }
      }
    }
#ifdef GEMM_STAMPS
    gs_epi += GS_NOW() - gs_t0;
#endif
    if (!more) break;
    __syncthreads();  // every wave is done with its transpose slice before the staging buffers are refilled
  }
#ifdef GEMM_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    long long *q = gemm_stamps + blockIdx.x * 8;
    q[1] = (long long)__builtin_amdgcn_s_memrealtime();
    q[2] += gs_loop;
    q[3] += gs_epi;
    q[4] += gs_tiles;
    q[5] = GS_NOW() - gs_mark;
  }
#endif
}

// MIXED: workgroups >= g.big_blocks compute 64x64 tiles of the rows [m_split, M) (same launch, same pipeline).
// PERSIST: gridDim.x resident workgroups walk the 128x128 tiles, then (MIXED) the 64x64 tiles of the remaining rows.
template <int WM, int WN, bool KTAIL, int NBUF, bool MIXED, bool PERSIST>
__global__ __launch_bounds__(256, (WM == 1 && WN == 1) ? 4 : (NBUF == 1 ? 3 : 2)) void gemm_f32_kernel(GemmArgs g) {
  constexpr int TBM = 64 * WM, TBN = 64 * WN;
  constexpr int EP_FLOATS = 4 * 32 * 32 * WN;  // epilogue transpose slices of the 4 waves (32 rows at a time)
  constexpr int ST_FLOATS = NBUF * (TBM + TBN) * LDS_LD;
  // one LDS array: [A buffers | W buffers]; the epilogue reuses it as 4 per-wave transpose slices
  __shared__ __attribute__((aligned(16))) float smem[ST_FLOATS > EP_FLOATS ? ST_FLOATS : EP_FLOATS];
  {
    const long bz = blockIdx.y;
    g.A += bz * g.sA;
    g.W += bz * g.sW;
    g.C += bz * g.sC;
    if (g.R) g.R += bz * g.sR;
  }
  if constexpr (PERSIST) {
    gemm_tile<WM, WN, KTAIL, NBUF, true>(g, (int)blockIdx.x, (int)gridDim.x, 0, g.m_split, smem);
    if constexpr (MIXED) {
      __syncthreads();
      // dealt from the far end: the workgroups with one big tile fewer take the small tiles first
      gemm_tile<1, 1, KTAIL, NBUF, true>(g, (int)(gridDim.x - 1 - blockIdx.x), (int)gridDim.x, g.m_split, g.M, smem);
    }
  } else {
    if constexpr (MIXED) {
      if ((int)blockIdx.x >= g.big_blocks) {
        gemm_tile<1, 1, KTAIL, NBUF, false>(g, (int)blockIdx.x - g.big_blocks, 0, g.m_split, g.M, smem);
        return;
      }
    }
    gemm_tile<WM, WN, KTAIL, NBUF, false>(g, (int)blockIdx.x, 0, 0, g.m_split, smem);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Latency variant for UNDER-FILLED launches (tile mode 15): single-image requests (M = 289 / 545 / 256 rows: the
// reference's own compress.py loop calls encode_only with B = 1), the CLIP towers, the bottleneck.
// With 64x64 tiles such a GEMM occupies a fraction of the chip (289 x 1024 -> 80 workgroups on 256 CUs) and every wave
// walks the tile's whole K range as ONE dependent MFMA chain: K x 32 cycles, 14 us at K = 1024, whatever the grid.
// Here a workgroup owns a 32x32 tile and each of its 4 waves a 16x16 block on v_mfma_f32_16x16x4_f32: 4 k per 32-cycle
// issue, so the chain is K x 10 cycles (40-cycle dependent latency) and there are 4x the workgroups to fill the chip.
// The k order is the one every other mode uses -- inside an 8-k group: 0,4,1,5 | 2,6,3,7 (lane group g = lane >> 4 of a
// 16x16x4 MFMA is accumulated g = 0..3; the LDS image of a slice is permuted so that each lane group reads its two k's as
// one float2) -- and the result is BITWISE the same (tools/micro/mfma16_order.hip; tests: all modes equal).
// Loads: K slices of 64 through a ring of three LDS buffers, requested three steps ahead and parked in two register
// sets (a step is only ~0.3 us of MFMA, far less than a load's latency).  The W fragment is the MFMA's A operand, so a
// lane ends up with 4 consecutive columns of one C row: the epilogue is a float4 store from registers.
// ------------------------------------------------------------------------------------------------------------------
#ifndef LAT_SETS
#define LAT_SETS 2   // K slices in flight per workgroup (register sets); 2..8 measured equal
#endif
#define LAT_BK 64
#define LAT_LD 68   // 64 + 4 floats: 16-byte aligned rows, ds_read_b128 nearly conflict-free (one 2-way slot per group)

template <int N, int I = 0, typename F>
__device__ __forceinline__ void lat_for(F &&f) {   // f(IntC<0>) ... f(IntC<N-1>): compile-time indices for the register sets
  if constexpr (I < N) {
    f(IntC<I>{});
    lat_for<N, I + 1>(f);
  }
}

template <int ACT, bool HASR>
__device__ __forceinline__ void lat_store(const GemmArgs &g, const f32x4 &acc, int m, int n) {
  if (m >= g.M || n >= g.N) return;
  f32x4 v = acc;
  if (g.bias) {
    const f32x4 b = *reinterpret_cast<const f32x4 *>(g.bias + n);
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] += b[t];
  }
#pragma unroll
  for (int t = 0; t < 4; t++) v[t] = apply_act_c<ACT>(v[t]);
  if constexpr (HASR) v += *reinterpret_cast<const f32x4 *>(g.R + (size_t)m * g.ldr + n);
  const size_t crow = g.c_seg ? (size_t)(m / g.c_seg) * g.c_seg_stride + (m % g.c_seg) : (size_t)m;
  *reinterpret_cast<f32x4 *>(g.C + crow * g.ldc + n) = v;
}

template <int R>   // R register sets: R K-slices (16 KB each per workgroup) in flight
__global__ __launch_bounds__(256, 2) void gemm_lat16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float sA[3][32 * LAT_LD];
  __shared__ __attribute__((aligned(16))) float sB[3][32 * LAT_LD];
  const int tiles_n = (g.N + 31) / 32;
  const int m0 = ((int)blockIdx.x / tiles_n) * 32, n0 = ((int)blockIdx.x % tiles_n) * 32;   // n fastest: neighbours share A rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 16, wn = (wave & 1) * 16;
  // staging map: a 32 x 64 slice is 512 float4 per operand: thread t moves float4 (row t >> 4, chunk t & 15) and row + 16
  const int srow = tid >> 4, schunk = tid & 15;
  const float *aptr[2], *wptr[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int am = min(m0 + srow + 16 * i, g.M - 1);
    const size_t arow = g.a_seg ? (size_t)(am / g.a_seg) * g.a_seg_stride + (am % g.a_seg) : (size_t)am;
    aptr[i] = g.A + arow * g.lda + schunk * 4;
    wptr[i] = g.W + (size_t)min(n0 + srow + 16 * i, g.N - 1) * g.ldw + schunk * 4;
  }
  f32x4 qa[R][2], qb[R][2];
  auto issue = [&](auto set_c, int k0) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      qa[S][i] = *reinterpret_cast<const f32x4 *>(aptr[i] + k0);
      qb[S][i] = *reinterpret_cast<const f32x4 *>(wptr[i] + k0);
    }
  };
  // LDS image of a slice row: every 8-k group is stored as [k0 k2 | k4 k6 | k1 k3 | k5 k7], so lane group grp of the 16x16x4
  // MFMAs finds ITS two k's (one for each of the group's two MFMAs) as one aligned float2 at offset 2 * grp -- no per-lane
  // element selection in the loop (it was 6 v_cndmask per MFMA), half the LDS read bytes, conflict-free ds_read_b64.
  // A thread stages the float4 of chunk c = k 4c .. 4c+3: elements (0,2) go to position 2 * (c & 1), elements (1,3) to 4 + 2 * (c & 1).
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int sgoff = (schunk >> 1) * 8 + 2 * (schunk & 1);
  auto store = [&](auto set_c, int buf) {
    constexpr int S = decltype(set_c)::value;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      float *pa = &sA[buf][(srow + 16 * i) * LAT_LD + sgoff], *pb = &sB[buf][(srow + 16 * i) * LAT_LD + sgoff];
      *reinterpret_cast<f32x2 *>(pa) = f32x2{qa[S][i][0], qa[S][i][2]};
      *reinterpret_cast<f32x2 *>(pa + 4) = f32x2{qa[S][i][1], qa[S][i][3]};
      *reinterpret_cast<f32x2 *>(pb) = f32x2{qb[S][i][0], qb[S][i][2]};
      *reinterpret_cast<f32x2 *>(pb + 4) = f32x2{qb[S][i][1], qb[S][i][3]};
    }
  };
  const int r16 = lane & 15, grp = lane >> 4;
  const int aoff = (wm + r16) * LAT_LD + 2 * grp, boff = (wn + r16) * LAT_LD + 2 * grp;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf) {   // 8 groups of 8 k: two MFMAs each, one dependent chain
    f32x2 fa[8], fb[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
      fa[s] = *reinterpret_cast<const f32x2 *>(&sA[buf][aoff + s * 8]);
      fb[s] = *reinterpret_cast<const f32x2 *>(&sB[buf][boff + s * 8]);
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s][0], fa[s][0], acc, 0, 0, 0);   // k = 0,4,1,5 by lane group
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[s][1], fa[s][1], acc, 0, 0, 0);   // k = 2,6,3,7
    }
  };
  const int nk = g.K / LAT_BK;
  // slice s is computed from LDS buffer s % 3; it was requested R + 1 steps earlier into register set s % R and written to
  // LDS one step before its use, so R slices are always in flight behind the one being computed
  issue(IntC<0>{}, 0);
  store(IntC<0>{}, 0);
  lat_for<R>([&](auto i_c) {   // slices 1 .. R (slice R goes into set 0, just freed)
    constexpr int I = decltype(i_c)::value + 1;
    if (I < nk) issue(IntC<I % R>{}, I * LAT_BK);
  });
  __syncthreads();
  int cur = 0;
  for (int base = 0; base < nk; base += R) {
    lat_for<R>([&](auto i_c) {
      constexpr int I = decltype(i_c)::value;
      const int kt = base + I;
      if (kt < nk) {                                   // uniform
        const int nxt = cur == 2 ? 0 : cur + 1;
        if (kt + 1 < nk) {
          store(IntC<(I + 1) % R>{}, nxt);              // slice kt + 1 lives in set (kt + 1) % R = (I + 1) % R (base % R == 0)
          if (kt + 1 + R < nk) issue(IntC<(I + 1) % R>{}, (kt + 1 + R) * LAT_BK);
        }
        compute(cur);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = nxt;
      }
    });
  }
  // accumulator: C[m = tile row + wm + r16][n = tile col + wn + 4 grp + 0..3]
  const int m = m0 + wm + r16, n = n0 + wn + 4 * grp;
  auto run_act = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    if (g.R) lat_store<ACT, true>(g, acc, m, n);
    else lat_store<ACT, false>(g, acc, m, n);
  };
  switch (g.act) {
    case ACT_GELU: run_act(IntC<ACT_GELU>{}); break;
    case ACT_SILU: run_act(IntC<ACT_SILU>{}); break;
    case ACT_TANH: run_act(IntC<ACT_TANH>{}); break;
    case ACT_LRELU: run_act(IntC<ACT_LRELU>{}); break;
    default: run_act(IntC<ACT_NONE>{}); break;
  }
}

// Launch options arrive per call (sgic_launch_opts, include/sgic.h): there is no process-global launch state.
// tile_mode: 0 = heuristic; 1 = 128x128 / 2 LDS buffers, 2 = 128x64 / 2 buffers, 3 = 128x128 / 1 buffer, 4 = 128x64 / 1 buffer,
// 5..8 = 1..4 with the start-up stagger, 9 = mixed 128x128 + 64x64 tail (2 buffers), 10 = mixed, 1 buffer,
// 11 = persistent 128x128 (2 buffers, 2 workgroups per CU walk all tiles), 12 = persistent mixed,
// 13 / 14 = 64x64 tiles as their own launch (2 / 1 buffers, 4+ workgroups per CU): small-M GEMMs (CLIP tower, bottleneck),
// 15 = latency kernel: 32x32 tiles of 16x16x4-MFMA blocks for under-filled launches (gemm_lat16_kernel)
#define SGIC_TILE_MODES 15

// Per-launch timing without extra packets on the stream: a launch that carries a profiler goes through
// hipExtLaunchKernel with its own (start, stop) event pair, i.e. the timestamps are taken by the dispatch itself
// (bench.py's live roofline figure).  Bracketing each launch with hipEventRecord instead costs ~2 us of stream time
// per event -- 1.5 % of a compress step at 360 launches.  One profiler per launching thread (not thread-safe itself).
extern "C" int sgic_profiler_create(sgic_profiler **out) {
  SGIC_REQUIRE(out, "out");
  *out = new (std::nothrow) sgic_profiler();
  SGIC_REQUIRE(*out, "out of memory");
  return SGIC_OK;
}

extern "C" void sgic_profiler_destroy(sgic_profiler *p) {
  if (!p) return;
  for (auto &e : p->pool) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  delete p;
}

extern "C" int sgic_profiler_begin(sgic_profiler *p, int max_launches) {
  SGIC_REQUIRE(p && max_launches > 0, "profiler / max_launches");
  while ((int)p->pool.size() < max_launches) {
    hipEvent_t a, b;
    SGIC_HIP(hipEventCreate(&a));
    SGIC_HIP(hipEventCreate(&b));
    p->pool.emplace_back(a, b);
  }
  p->n = 0;
  return SGIC_OK;
}

// closes the window; ms_out[i] = duration of the i-th launch since sgic_profiler_begin (the call synchronises on each
// stop event)
extern "C" int sgic_profiler_end(sgic_profiler *p, float *ms_out, int cap, int *n_out) {
  SGIC_REQUIRE(p && p->n >= 0 && n_out, "no open profile window");
  const int total = p->n, n = total < cap ? total : cap;   // total <= pool size: prof_next grows the pool first
  p->n = -1;
  for (int i = 0; i < n; i++) {
    SGIC_HIP(hipEventSynchronize(p->pool[i].second));
    SGIC_HIP(hipEventElapsedTime(&ms_out[i], p->pool[i].first, p->pool[i].second));
  }
  *n_out = total;  // launches seen in the window (the caller compares it with its own count)
  return SGIC_OK;
}

template <typename K>
static inline void launch_gemm(K kernel, dim3 grid, hipStream_t st, const GemmArgs &g, const sgic_launch_opts *o) {
  if (const auto *evp = prof_next(o)) {
    const auto &ev = *evp;
    hipExtLaunchKernelGGL(kernel, grid, dim3(256), 0, st, ev.first, ev.second, 0, g);
  } else {
    kernel<<<grid, 256, 0, st>>>(g);
  }
}

static int gemm_launch(GemmArgs g, int batch, hipStream_t st, const sgic_launch_opts *o) {
  const int M = g.M, N = g.N, K = g.K;
  // Tile choice: 128x128 unless the 128x64 grid fills the last round of workgroups on the 256 CUs clearly
  // better (workgroups are dispatched dynamically, so the makespan is ~ceil(blocks / 256) block-times).
  const int tm128 = (M + 127) / 128;
  auto eff = [&](int bn) {
    const double nb = (double)tm128 * ((N + bn - 1) / bn) * batch / 256.0;
    return nb / ceil(nb);
  };
  bool narrow = N <= 64 || eff(64) > eff(128) + 0.04;
  bool single = false, mixed = false, persist = false, tiny = false;
  int tile_mode = o ? o->tile_mode : 0;
  SGIC_REQUIRE(tile_mode >= 0 && tile_mode <= SGIC_TILE_MODES, "tile_mode");
  const bool stagger = tile_mode >= 5 && tile_mode <= 8;
  if (stagger) tile_mode -= 4;
  if (tile_mode) {
    narrow = (tile_mode == 2 || tile_mode == 4);
    single = (tile_mode == 3 || tile_mode == 4 || tile_mode == 10 || tile_mode == 14);
    mixed = tile_mode == 9 || tile_mode == 10 || tile_mode == 12;
    persist = (tile_mode == 11 || tile_mode == 12) && batch == 1;
    tiny = tile_mode >= 13;
  }
  const bool ktail = (K % BK) != 0;
  // mode 15 / heuristic: the latency kernel, when the launch under-fills the chip even with 64x64 tiles (measured with C++
  // launches, 64x64 kernel -> this one: 256x768x768 18.9 -> 11.4 us, 2048x128x256 8.5 -> 5.7 us, one 64x64x512 tile 12.9 ->
  // 8.0 us, 289x1024x1024 24.5 -> 23.0 us; slower from ~240 workgroups of 64x64 up) and the operands meet its (float4,
  // K % 64) requirements; anything else falls through to the ordinary modes
  {
    const bool lat_ok = batch == 1 && !g.conv_C && (K % LAT_BK) == 0 && g.vec_epilogue && (long)((M + 31) / 32) * ((N + 31) / 32) < (1 << 20);
    const long wg64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (lat_ok && (tile_mode == 15 || (!tile_mode && wg64 <= 96))) {
      const dim3 lgrid((unsigned)(((M + 31) / 32) * ((N + 31) / 32)));
      if (const auto *evp = prof_next(o)) {
        const auto &ev = *evp;
        hipExtLaunchKernelGGL(gemm_lat16_kernel<LAT_SETS>, lgrid, dim3(256), 0, st, ev.first, ev.second, 0, g);
      } else {
        gemm_lat16_kernel<LAT_SETS><<<lgrid, 256, 0, st>>>(g);
      }
      return sgic::check_launch("gemm_lat16_kernel");
    }
    if (tile_mode == 15) tile_mode = 13, tiny = true;   // not eligible: the 64x64 launch
  }
  // heuristic for under-filled grids: when the 128-row tiling leaves more than half of the 256 CUs without a workgroup,
  // 64x64 tiles give 4x the workgroups and a 4x shorter serial MFMA chain per wave (the latency of small-M GEMMs)
  if (!tile_mode && !ktail && batch == 1 && (long)tm128 * ((N + (narrow ? 63 : 127)) / (narrow ? 64 : 128)) < 128) tiny = true;
  if (ktail) single = mixed = persist = tiny = false;
  const int tn = narrow ? (N + 63) / 64 : (N + 127) / 128;
  g.m_split = M;
  g.big_blocks = tm128 * tn;
  int small_blocks = 0;
  if (mixed && batch == 1 && !narrow) {
    // 128x128 tiles for as many m-tile rows as fill whole multiples of the 256 CUs; the remaining rows (less than
    // one CU-round of work) go to 64x64 tiles, whose finer granularity ends the launch evenly
    const int round = persist ? 512 : 256;                    // resident workgroups (persistent) / CUs
    const int full = ((M / 128) * tn / round) * round;        // big tiles in whole rounds
    const int big_rows = (full / tn) * 128;
    if (big_rows <= 0 || big_rows >= M) {
      mixed = false;
    } else {
      g.m_split = big_rows;
      g.big_blocks = (big_rows / 128) * tn;
      small_blocks = ((M - big_rows + 63) / 64) * ((N + 63) / 64);
    }
  } else {
    mixed = false;
  }
  dim3 grid(g.big_blocks + small_blocks, batch);
  if (persist) grid.x = min((int)grid.x, 2 * 256);  // 2 resident workgroups per CU (LDS-bound), 256 CUs
  {
    const int per_cu = single ? (narrow ? 4 : 3) : 2;
    const long total = (long)g.big_blocks * batch;
    g.per_cu = per_cu;
    // one tile's MFMA time alone on a CU ~ nk * 64(32 narrow) MFMAs * 64 cycles; delay resident slot s by s/per_cu of it
    const long tile_cycles = (long)((K + BK - 1) / BK) * (narrow ? 32 : 64) * 64;
    g.stagger_cycles = (stagger && !mixed && total >= 2L * per_cu * 256) ? (int)(tile_cycles / per_cu) : 0;
  }
  if (tiny) {
    grid.x = ((M + 63) / 64) * ((N + 63) / 64);
    g.big_blocks = (int)grid.x;
    g.stagger_cycles = 0;
    if (single) launch_gemm(gemm_f32_kernel<1, 1, false, 1, false, false>, grid, st, g, o);
    else launch_gemm(gemm_f32_kernel<1, 1, false, 2, false, false>, grid, st, g, o);
  } else if (persist) {
    if (mixed) launch_gemm(gemm_f32_kernel<2, 2, false, 2, true, true>, grid, st, g, o);
    else launch_gemm(gemm_f32_kernel<2, 2, false, 2, false, true>, grid, st, g, o);
  } else if (mixed) {
    if (single) launch_gemm(gemm_f32_kernel<2, 2, false, 1, true, false>, grid, st, g, o);
    else launch_gemm(gemm_f32_kernel<2, 2, false, 2, true, false>, grid, st, g, o);
  } else if (narrow) {
    if (ktail) launch_gemm(gemm_f32_kernel<2, 1, true, 2, false, false>, grid, st, g, o);
    else if (single) launch_gemm(gemm_f32_kernel<2, 1, false, 1, false, false>, grid, st, g, o);
    else launch_gemm(gemm_f32_kernel<2, 1, false, 2, false, false>, grid, st, g, o);
  } else {
    if (ktail) launch_gemm(gemm_f32_kernel<2, 2, true, 2, false, false>, grid, st, g, o);
    else if (single) launch_gemm(gemm_f32_kernel<2, 2, false, 1, false, false>, grid, st, g, o);
    else launch_gemm(gemm_f32_kernel<2, 2, false, 2, false, false>, grid, st, g, o);
  }
  return sgic::check_launch("gemm_f32_kernel");
}

static int gemm_check(const float *d_A, int lda, const float *d_W, int ldw, const float *d_bias, const float *d_R, int ldr,
                      float *d_C, int ldc, int M, int N, int K, int act) {
  SGIC_REQUIRE(d_A && d_W && d_C && M > 0 && N > 0 && K > 0, "null/empty");
  SGIC_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0, "K, lda, ldw must be multiples of 4 floats");
  SGIC_REQUIRE(ldw >= K && ldc >= N && (!d_R || ldr >= N), "leading dimensions");
  SGIC_REQUIRE(((uintptr_t)d_A & 15) == 0 && ((uintptr_t)d_W & 15) == 0, "A and W must be 16-byte aligned");
  SGIC_REQUIRE(act >= 0 && act <= ACT_LRELU, "activation");
  (void)d_bias;
  return SGIC_OK;
}

static int vec_ok(const float *d_bias, const float *d_R, int ldr, float *d_C, int ldc, int N, long sC, long sR) {
  return (N % 4 == 0) && (ldc % 4 == 0) && (((uintptr_t)d_C) & 15) == 0 && (sC % 4 == 0) &&
         (!d_R || ((ldr % 4 == 0) && (((uintptr_t)d_R) & 15) == 0 && (sR % 4 == 0))) && (!d_bias || (((uintptr_t)d_bias) & 15) == 0);
}

// C[M,N] = act(A[M,K] @ W[N,K]^T + bias) + R      (nn.Linear / 1x1 conv semantics, all fp32)
extern "C" int sgic_gemm_f32(const float *d_A, int lda, const float *d_W, int ldw, const float *d_bias,
                             const float *d_R, int ldr, float *d_C, int ldc, int M, int N, int K, int act,
                             int a_seg, int a_seg_stride, int c_seg, int c_seg_stride, const sgic_launch_opts *opts,
                             sgic_stream_t stream) {
  int rc = gemm_check(d_A, lda, d_W, ldw, d_bias, d_R, ldr, d_C, ldc, M, N, K, act);
  if (rc) return rc;
  SGIC_REQUIRE(lda >= K, "lda");
  SGIC_REQUIRE(a_seg >= 0 && c_seg >= 0 && (a_seg == 0 || a_seg_stride >= a_seg) && (c_seg == 0 || c_seg_stride >= c_seg),
               "row segment maps");
  GemmArgs g{d_A, d_W, d_bias, d_R, d_C, M, N, K, lda, ldw, ldr, ldc, act, a_seg, a_seg_stride, c_seg, c_seg_stride,
             vec_ok(d_bias, d_R, ldr, d_C, ldc, N, 0, 0), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return gemm_launch(g, 1, to_stream(stream), opts);
}

// batch of independent GEMMs with element strides (stride 0 = operand shared by all batches); used for the
// single-head attention of the VQGAN AttnBlock (model.py:168-192): S_b = Q_b K_b^T, O_b = P_b V_b.
extern "C" int sgic_gemm_batched_f32(const float *d_A, int lda, long strideA, const float *d_W, int ldw, long strideW,
                                     const float *d_bias, const float *d_R, int ldr, long strideR, float *d_C, int ldc,
                                     long strideC, int M, int N, int K, int act, int batch, const sgic_launch_opts *opts,
                                     sgic_stream_t stream) {
  int rc = gemm_check(d_A, lda, d_W, ldw, d_bias, d_R, ldr, d_C, ldc, M, N, K, act);
  if (rc) return rc;
  SGIC_REQUIRE(lda >= K && batch > 0 && batch < 65536 && (strideA & 3) == 0 && (strideW & 3) == 0, "batch/strides");
  GemmArgs g{d_A, d_W, d_bias, d_R, d_C, M, N, K, lda, ldw, ldr, ldc, act, 0, 0, 0, 0,
             vec_ok(d_bias, d_R, ldr, d_C, ldc, N, strideC, strideR), strideA, strideW, strideC, strideR, 0, 0, 0, 0, 0, 0, 0};
  return gemm_launch(g, batch, to_stream(stream), opts);
}

// Thin-output 3x3 convolution (Cout <= 4, Cin == 128: the taming decoder's conv_out 128 -> 3, model.py:531-537).
// A 64-column MFMA tile would spend 95 % of its work on padding, so this one is a VALU kernel bound by reading the
// input once: 32 lanes share one pixel (lane = 4 channels, coalesced 512-byte rows per tap), each lane keeps its
// 9 x COUT weight float4s in registers, partial dot products are combined by a fixed xor-shuffle tree.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_thin_kernel(const float *__restrict__ in, const float *__restrict__ W,
                                                           const float *__restrict__ bias, float *__restrict__ out, int ldc,
                                                           long npix, int H, int Wd, int act) {
  constexpr int CIN = 128;
  const int l32 = threadIdx.x & 31;
  const long group = ((long)blockIdx.x * 256 + threadIdx.x) >> 5, ngroups = (long)gridDim.x * 8;
  f32x4 wr[COUT][9];
#pragma unroll
  for (int co = 0; co < COUT; co++)
#pragma unroll
    for (int tap = 0; tap < 9; tap++) wr[co][tap] = *reinterpret_cast<const f32x4 *>(W + ((size_t)co * 9 + tap) * CIN + l32 * 4);
  const int hw = H * Wd;
  for (long pix = group; pix < npix; pix += ngroups) {
    const int b = (int)(pix / hw), r = (int)(pix - (long)b * hw), y = r / Wd, x = r - y * Wd;
    const float *base = in + (((size_t)b * (H + 2) + y) * (Wd + 2) + x) * CIN + l32 * 4;
    f32x4 v[9];
#pragma unroll
    for (int tap = 0; tap < 9; tap++) v[tap] = *reinterpret_cast<const f32x4 *>(base + (size_t)((tap / 3) * (Wd + 2) + tap % 3) * CIN);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) {
      float a = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int t = 0; t < 4; t++) a = fmaf(v[tap][t], wr[co][tap][t], a);
      acc[co] = a;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int co = 0; co < COUT; co++) acc[co] += __shfl_xor(acc[co], o);
    if (l32 < COUT) {
      float r0 = acc[0];
#pragma unroll
      for (int co = 1; co < COUT; co++) r0 = (l32 == co) ? acc[co] : r0;
      out[(size_t)pix * ldc + l32] = apply_act(r0 + (bias ? bias[l32] : 0.f), act);
    }
  }
}

// 3x3 stride-1 pad-1 convolution as an implicit GEMM over a zero-halo NHWC input [B, H+2, W+2, Cin]:
// out[(b,y,x), n] = act(sum_{ky,kx,c} in[b, y+ky, x+kx, c] * W[n, (ky*3+kx)*Cin + c] + bias[n]) + R
extern "C" int sgic_conv3x3_f32(const float *d_in_halo, const float *d_W, const float *d_bias, const float *d_R, int ldr,
                                float *d_out, int ldc, int B, int H, int W, int Cin, int Cout, int act,
                                const sgic_launch_opts *opts, sgic_stream_t stream) {
  const long Ml = (long)B * H * W;
  SGIC_REQUIRE(Ml < (1l << 31), "too many pixels");
  const int M = (int)Ml, K = 9 * Cin;
  int rc = gemm_check(d_in_halo, Cin, d_W, K, d_bias, d_R, ldr, d_out, ldc, M, Cout, K, act);
  if (rc) return rc;
  SGIC_REQUIRE(Cin % BK == 0, "implicit-GEMM conv needs Cin % 32 == 0");
  if (Cout == 3 && Cin == 128 && !d_R) {
    const unsigned grid = (unsigned)min((Ml + 7) / 8, 256L * 32);
    if (const auto *evp = prof_next(opts)) {
      const auto &ev = *evp;
      hipExtLaunchKernelGGL(conv3x3_thin_kernel<3>, dim3(grid), dim3(256), 0, to_stream(stream), ev.first, ev.second, 0,
                            d_in_halo, d_W, d_bias, d_out, ldc, Ml, H, W, act);
    } else {
      conv3x3_thin_kernel<3><<<grid, 256, 0, to_stream(stream)>>>(d_in_halo, d_W, d_bias, d_out, ldc, Ml, H, W, act);
    }
    return sgic::check_launch("conv3x3_thin_kernel");
  }
  GemmArgs g{d_in_halo, d_W, d_bias, d_R, d_out, M, Cout, K, Cin, K, ldr, ldc, act, 0, 0, 0, 0,
             vec_ok(d_bias, d_R, ldr, d_out, ldc, Cout, 0, 0), 0, 0, 0, 0, Cin, H, W, 0, 0, 0, 0};
  return gemm_launch(g, 1, to_stream(stream), opts);
}
