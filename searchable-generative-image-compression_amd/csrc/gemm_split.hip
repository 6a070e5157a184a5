// fp32-accurate GEMM on the bf16 matrix pipe ("bf16x3" operand split):  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// The fp32-input MFMA (gemm.hip) runs at 1/16 of the bf16 MFMA rate on CDNA4 and gfx950 has no xf32/TF32 mode.  Here
// each fp32 operand is written as the exact sum of three bf16 values,  x = x1 + x2 + x3  (x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2); 3 x 8 significant bits carry all 24 bits of an fp32), and a product as
//     a.w  ~=  a1 w1 + (a1 w2 + a2 w1) + (a2 w2 + a1 w3 + a3 w1),
// six v_mfma_f32_16x16x32_bf16 (fp32 accumulate; each bf16 x bf16 product is exact in fp32) per 16x16 block and 32 k instead
// of 2 x 8 fp32 MFMAs: 6/16 of the matrix-pipe time.  The dropped terms (a2 w3, a3 w2, a3 w3) are <= 2^-26 of the product, a
// quarter of an fp32 ulp.  Measured against an fp64 reference (tools/micro/split3_gemm.hip, K = 4096, random normal
// data): rms error 1.01e-6 of rms(C) vs 1.14e-6 for the plain fp32 fmaf chain that gemm.hip (and any fp32 GEMM)
// computes -- this is an fp32 GEMM in accuracy, not a reduced-precision one; tests/test_gpu_split3.py holds that bound.
//
// Per output element the arithmetic is a fixed sequence fixed by K alone (k slices of 32 in order, six MFMAs per slice in the
// term order a3w1, a1w3, a2w2, a2w1, a1w2, a1w1): no split-K, and the same for every tile shape below, so results are bitwise
// independent of M, of the tile choice and of the batch a row sits in (batch-invariant), like gemm.hip.
//
// Operands arrive PRE-SPLIT as three bf16 planes -- slice-major [3][K / 32][rows][32] (split3.h; round 3) or row-major [3][rows][K],
// per operand (S3Args.a_packed / w_packed): weights once at load time, activations by
// split3_rows_kernel (or directly by the producing kernel).  The GEMM kernel is then a pure bf16 pipeline:
//  * tile (16 BM WAVES_M) x (16 BN WAVES_N), K slices of 32, two LDS stages; rows are 64 B (32 k) unpadded, 16-byte slots
//    XOR-swizzled per 4-row group: conflict-free ds_read_b128 / ds_write_b128;
//  * global -> VGPR -> LDS staging one slice ahead (loads issued two slices ahead), one barrier per slice placed where the
//    remaining MFMAs cover the first LDS reads of the next slice (see the kernel);
//  * XCD-aware grouped tile walk as in gemm.hip;
//  * the 16x16x32 shape (rather than 32x32x16) keeps the dependent MFMA chain of a block at 6 x 16 cycles per 32 k, which is
//    what the 32x32-tile latency variant for single-image requests lives on, and holds a higher clock under load.
#include <type_traits>

#include "common.h"
#include "gemm_act.h"
#include "profiler.h"
#include "split3.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x[rows, cols] fp32 (row map as sgic_gemm_f32's A operand) -> planes [3][rows][cols] bf16; 8 elements per thread
// PACK: the slice-major layout [3][cols / 32][rows][32] instead (a constant W operand: 16 rows of a 32-k slice = 1 KiB contiguous)
template <bool PACK>
__global__ __launch_bounds__(256) void split3_rows_kernel(const float *__restrict__ x, int ld, int rows, int cols, int seg,
                                                          int seg_stride, unsigned short *__restrict__ planes) {
  const int c8 = cols >> 3;
  const long total = (long)rows * c8, plane = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / c8), c = (int)(i - (long)r * c8) * 8;
    const size_t src = (seg ? (size_t)(r / seg) * seg_stride + (r % seg) : (size_t)r) * ld + c;
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(x + src), v1 = *reinterpret_cast<const f32x4 *>(x + src + 4);
    unsigned p1[4], p2[4], p3[4];
    s3_split_pair(v0[0], v0[1], p1[0], p2[0], p3[0]);
    s3_split_pair(v0[2], v0[3], p1[1], p2[1], p3[1]);
    s3_split_pair(v1[0], v1[1], p1[2], p2[2], p3[2]);
    s3_split_pair(v1[2], v1[3], p1[3], p2[3], p3[3]);
    const size_t dst = PACK ? ((size_t)(c >> 5) * rows + r) * 32 + (c & 31) : (size_t)r * cols + c;
    *reinterpret_cast<u32x4 *>(planes + dst) = u32x4{p1[0], p1[1], p1[2], p1[3]};
    *reinterpret_cast<u32x4 *>(planes + plane + dst) = u32x4{p2[0], p2[1], p2[2], p2[3]};
    *reinterpret_cast<u32x4 *>(planes + 2 * plane + dst) = u32x4{p3[0], p3[1], p3[2], p3[3]};
  }
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void s3_for(F &&f) {   // f(IntC<0>) ... f(IntC<N-1>): compile-time indices for the register sets
  if constexpr (I < N) {
    f(IntC<I>{});
    s3_for<N, I + 1>(f);
  }
}

struct S3Args {
  const unsigned short *A, *W;   // planes [3][M][K], [3][N][K] (row-major) or slice-major: a_packed / w_packed below
  const float *bias, *R;
  float *C;
  int M, N, K, ldr, ldc, act;
  int c_seg, c_seg_stride;
  long a_plane, w_plane;         // elements between planes
  unsigned short *Cp;            // optional: the result as bf16x3 planes [3][M][N] (the next GEMM's A operand) instead of C
  long c_plane;                  // elements between the planes of Cp
  int vec_epilogue;              // C / R / bias rows are float4-addressable
  // implicit-GEMM 3x3 convolution (stride 1, pad 1), as gemm.hip: A = planes of a zero-halo NHWC buffer [B, H+2, W+2, C], logical
  // row m = output pixel (b,y,x), K = 9 C ordered (ky,kx,c); conv_C == 0 disables.  C % 32 == 0: a K slice never straddles a tap
  int conv_C, conv_H, conv_W;
  int m_base;                    // logical row of this launch's row 0 (second launch of the split modes 6 / 7): enters the C row map
  int w_packed;                  // W planes are slice-major [3][K / 32][N][32] (sgic_split3_pack_f32) instead of [3][N][K]
  int a_packed, a_rows;          // A planes are slice-major [3][K / 32][a_rows][32] (split3.h); convolution: [3][conv_C / 32][a_rows = halo rows][32]
  int c_rows;                    // rows of the Cp planes (slice-major [3][N / 32][c_rows][32]): the whole product's M, also in a split launch
};

// ---- epilogue of one tile, shared by the register-staged and the LDS-DMA kernel: the accumulators hold
// C[m = block row base + l16][n = block column base + 4 lq .. + 3].  bias -> activation -> residual, as gemm.hip.  Wide path: the wave
// re-distributes its tile through its own LDS slice `ep` (row stride + 4 floats: conflict-free ds_write_b128) so that a store
// instruction writes whole row segments (16 BN floats per row), as float4 or -- when the consumer is the next GEMM -- directly as
// bf16x3 planes.  PASSES: the tile goes through the slice in that many groups of block rows (1: the slice holds the wave's whole
// tile, 16 BM rows; 2: half of it at a time -- the LDS-DMA kernel, whose other staging buffer is being filled meanwhile).
template <int BM, int BN, int PASSES>
__device__ __forceinline__ void s3_tile_epilogue(const S3Args &g, f32x4 (&acc)[BM][BN], float *ep, int em0, int en0, int wm, int wn, int lane) {
  const int l16 = lane & 15, lq = lane >> 4;
  if (g.vec_epilogue) {
    // LPR lanes cover a row segment of the wave's tile, RPI rows per store instruction.  BN = 3 (the 128x192 tile: 48 columns per wave):
    // 12 lanes per row, five rows per instruction on 60 lanes, seven instructions for 32 rows with the last one partial (EXACT = false)
    constexpr int BMP = BM / PASSES, RM = 16 * BMP, CN = 16 * BN, LDW = CN + 4, LPR = CN / 4, RPI = 64 / LPR, NIT = (RM + RPI - 1) / RPI;
    constexpr bool EXACT = (64 % LPR) == 0 && (RM % RPI) == 0;
    static_assert(BM % PASSES == 0, "epilogue passes");
    const int c4 = (lane % LPR) * 4, r0 = lane / LPR;
    const bool lane_on = EXACT || lane < RPI * LPR;
    const int n = en0 + wn + c4;
    const bool colok = n < g.N && lane_on;
    const int nc = colok ? n : 0;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && colok) bv = *reinterpret_cast<const f32x4 *>(g.bias + n);
    s3_for<PASSES>([&](auto pass_c) __attribute__((always_inline)) {   // compile-time pass index: acc[][] stays in registers
      constexpr int pass = decltype(pass_c)::value;
#pragma unroll
      for (int i = 0; i < BMP; i++)
#pragma unroll
        for (int j = 0; j < BN; j++) *reinterpret_cast<f32x4 *>(ep + (i * 16 + l16) * LDW + j * 16 + 4 * lq) = acc[pass * BMP + i][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int wmp = wm + pass * RM;
      auto run = [&](auto act_c, auto res_c, auto pl_c) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_c)::value;
        constexpr bool HASR = decltype(res_c)::value != 0, PLANES = decltype(pl_c)::value != 0;
        constexpr int EB = NIT < 8 ? NIT : 8;    // rows in flight between the LDS read / residual load and the store
#pragma unroll
        for (int b = 0; b < NIT; b += EB) {
          f32x4 cv[EB], rv[EB];
#pragma unroll
          for (int q = 0; q < EB; ++q) {
            const int rr = EXACT ? (b + q) * RPI + r0 : min((b + q) * RPI + r0, RM - 1);   // (a clamped row is not stored: see below)
            cv[q] = *reinterpret_cast<const f32x4 *>(ep + rr * LDW + c4);
            if constexpr (HASR) rv[q] = *reinterpret_cast<const f32x4 *>(g.R + (size_t)(min(em0 + wmp + rr, g.M - 1) + g.m_base) * g.ldr + nc);
          }
#pragma unroll
          for (int q = 0; q < EB; ++q) {
            if (!EXACT && (b + q >= NIT || (b + q) * RPI + r0 >= RM)) continue;
            const int m = em0 + wmp + (b + q) * RPI + r0;
            f32x4 v = cv[q];
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = apply_act_c<ACT>(v[t] + bv[t]);
            if constexpr (HASR) v += rv[q];
            if (colok && m < g.M) {
              if constexpr (PLANES) {
                s3_store4(g.Cp, g.c_plane, s3_pack_off(m + g.m_base, n, g.c_rows), v);
              } else {
                const int mm = m + g.m_base;
                const size_t crow = g.c_seg ? (size_t)(mm / g.c_seg) * g.c_seg_stride + (mm % g.c_seg) : (size_t)mm;
                *reinterpret_cast<f32x4 *>(g.C + crow * g.ldc + n) = v;
              }
            }
          }
        }
      };
      auto run_act = [&](auto act_c) __attribute__((always_inline)) {
        if (g.Cp) {
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
    });
    return;
  }
  // narrow path (N or the leading dimensions not float4-addressable): one dword at a time from the accumulators
  auto run = [&](auto act_c, auto res_c) __attribute__((always_inline)) {
    constexpr int ACT = decltype(act_c)::value;
    constexpr bool HASR = decltype(res_c)::value != 0;
#pragma unroll
    for (int j = 0; j < BN; j++) {
      const int n = en0 + wn + j * 16 + 4 * lq;
#pragma unroll
      for (int i = 0; i < BM; i++) {
        const int m = em0 + wm + i * 16 + l16;
        if (m >= g.M) continue;
        const int mm = m + g.m_base;
        const size_t crow = g.c_seg ? (size_t)(mm / g.c_seg) * g.c_seg_stride + (mm % g.c_seg) : (size_t)mm;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (n + t < g.N) {
            float o = apply_act_c<ACT>(acc[i][j][t] + (g.bias ? g.bias[n + t] : 0.f));
            if constexpr (HASR) o += g.R[(size_t)mm * g.ldr + n + t];
            g.C[crow * g.ldc + n + t] = o;
          }
        }
      }
    }
  };
  auto run_act = [&](auto act_c) __attribute__((always_inline)) {
    if (g.R) run(act_c, IntC<1>{});
    else run(act_c, IntC<0>{});
  };
  switch (g.act) {
    case ACT_GELU: run_act(IntC<ACT_GELU>{}); break;
    case ACT_SILU: run_act(IntC<ACT_SILU>{}); break;
    case ACT_TANH: run_act(IntC<ACT_TANH>{}); break;
    case ACT_LRELU: run_act(IntC<ACT_LRELU>{}); break;
    default: run_act(IntC<ACT_NONE>{}); break;
  }
}

// Tile = (16 BM WAVES_M) x (16 BN WAVES_N); every wave owns BM x BN blocks of 16 x 16 on v_mfma_f32_16x16x32_bf16 (one MFMA =
// one plane pair over a whole 32-k slice).  The W fragment is passed as the MFMA's first operand, so a lane ends up with 4
// CONSECUTIVE columns of one C row (row = lane & 15, columns 4 (lane >> 4) ..): the epilogue stores float4 / 8-byte plane
// pieces straight from the accumulators, no transpose.
//
// Schedule of one K slice (BN >= 2; W fragments = two halves Wlo / Whi of BN/2 column blocks, held for the whole slice; A
// fragments per block row, double-buffered):
//     [ds_write slice kt+1 -> other LDS stage][global loads of slice kt+2]
//     row 0:  Whi blocks (prefetched during the previous slice's tail), meanwhile read Wlo;  then Wlo blocks
//     rows 1 .. BM-2: all blocks (A of the next row prefetched one row ahead)
//     row BM-1: Whi blocks | BARRIER | read Whi and A row 0 of slice kt+1 from the other stage | Wlo blocks
// i.e. the one barrier per slice sits 6 BN/2 MFMAs before the slice's end, where every LDS read of the current stage has been
// issued, and the MFMAs left cover the latency of the first reads of the next stage: the matrix pipe never waits at the seam
// and no fragment is held twice (acc 4 BM BN + W 12 BN + A 24 registers).
// NS = register sets of staged K stages: 1 = loads run two stages ahead (large tiles: a stage is >= 1536 MFMA cycles, longer than a
// load); NS > 1 = the loads of stage kt + 1 + NS are issued at stage kt (small tiles of under-filled launches: a stage is a few
// hundred MFMA cycles and one workgroup per CU has nothing else to hide a ~2 us load behind).
// KS = 32-k slices per LDS stage (rows of 64 KS bytes): 2 for the small tiles when K % 64 == 0 -- one barrier and one round of
// exposed LDS latency per 64 k instead of per 32 k, which is what bounds a launch of one wave per SIMD.
// PERSIST: gridDim.x resident workgroups (one per CU) walk the tile list; the next tile's first K stage is requested BEFORE the
// current tile's epilogue, so the first-load latency of a tile hides behind the previous tile's stores.
// Diagnostic build only (tools/micro/launch_latency.hip defines S3_STAMPS): s_memrealtime (100 MHz) of the first workgroup's
// start and the last workgroup's end, and s_memtime (shader clock) over the same span of workgroup 0 -- what a launch takes
// on the shader array itself, beside its dispatch timestamps.  No stamp executes in the product build.
// S3_ABLATE (diagnostic builds of tools/micro/split3_phases.hip only; results are garbage): bit 0 = no global loads in the main
// loop, bit 1 = no LDS writes, bit 2 = no barrier, bit 3 = no fragment reads -- what each part of the K loop costs beside the MFMAs;
// bits 4 / 5 / 6 (LDS-DMA kernel): operand reads of tile (0,0) / (m,0) only; operands addressed as if stored tile-packed (6+7: W only);
// bit 8: the third plane of both operands is not re-fetched after the first slice (a third fewer operand bytes per slice)
#ifndef S3_ABLATE
#define S3_ABLATE 0
#endif
#ifndef S3_DMA_VARIANT
#define S3_DMA_VARIANT 0   // diagnostic builds: bit 0 = no sched_group_barrier in the DMA kernel's slice, bit 1 = s_setprio(1) on the younger half of the waves
#endif
#ifdef S3_STAMPS
__device__ unsigned long long s3_stamps[4] = {~0ull, 0ull, 0ull, 0ull};   // min start, max end (realtime); wg 0: start, end (s_memtime)
// per workgroup (wave 0): shader cycles summed over its tiles in {prologue, main loop, epilogue}, tiles, realtime start / end
__device__ unsigned long long s3_wg[1024][6];
#endif

template <int WAVES_M, int WAVES_N, int BM, int BN, int NS, int KS, bool PERSIST>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, (2 * 3 * 16 * (BM * WAVES_M + BN * WAVES_N) * 64 * KS <= 80 * 1024 && WAVES_M * WAVES_N == 8) ? 2 : 1)
void gemm_split3_kernel(S3Args g) {
#ifdef S3_STAMPS
  const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_c0 = __builtin_amdgcn_s_memtime();
#endif
  static_assert(KS == 1 || KS == 2, "32-k slices per stage");
  static_assert(!PERSIST || NS == 1, "the persistent walk is built on the one-register-set pipeline");
  constexpr int NT = 64 * WAVES_M * WAVES_N, TM = 16 * BM * WAVES_M, TN = 16 * BN * WAVES_N;
  constexpr int ROWB = 64 * KS, SL = 4 * KS;                                    // bytes / 16-byte slots per LDS row
  constexpr int CA = (TM * SL + NT - 1) / NT, CW = (TN * SL + NT - 1) / NT;     // 16-byte chunks per thread, plane and stage
  constexpr int APLANE = TM * ROWB, WPLANE = TN * ROWB, STAGE = 3 * (APLANE + WPLANE);
  constexpr int HB = BN >= 2 ? BN / 2 : BN;     // column blocks in the "hi" half of W (BN == 1: no halves)
  static_assert((NT / SL) % 16 == 0, "a thread's staged rows must share their swizzle term");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN, nwg = tiles_m * tiles_n;
  int m0, n0;
  auto locate = [&](int t) __attribute__((always_inline)) {   // XCD-aware bijective remap + grouped walk (8 m-tiles x all n-tiles, m fastest), as gemm.hip
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, within = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    const int per_group = 8 * tiles_n, group = t / per_group, first_m = group * 8, gsz = min(tiles_m - first_m, 8),
              in_group = t - group * per_group;
    m0 = (first_m + in_group % gsz) * TM;
    n0 = (in_group / gsz) * TN;
  };
  int tile = blockIdx.x;
  locate(tile);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WAVES_N) * (16 * BM), wn = (wave % WAVES_N) * (16 * BN);
  // LDS image of a stage: per plane, rows of 64 KS bytes unpadded; the 16-byte slot s of row r sits at slot s ^ swz(r):
  //   KS = 1: swz = f((r >> 2) & 3), f = (0, 2, 3, 1);   KS = 2: swz = (r >> 1) & 7
  // -- conflict-free for the ds_read_b128 lane groups {0-3,12-15,20-27}, ... of a fragment read (row = lane & 15,
  // slot = 4 slice + (lane >> 4)) and for ds_write_b128 (8 consecutive lanes = whole rows)
  auto swz = [](int row) __attribute__((always_inline)) {
    if constexpr (KS == 1) {
      const int gq = (row >> 2) & 3;
      return (((gq ^ (gq >> 1)) & 1) << 1) | (gq >> 1);
    } else {
      return (row >> 1) & 7;
    }
  };
  // staging map: chunk q = tid + i NT of a plane -> (row q / SL, slot q % SL); NT / SL is a multiple of 16, so the swizzle of a
  // thread's rows does not depend on i
  const int srow = tid / SL, sslot = tid % SL;
  const int sw = srow * ROWB + ((sslot ^ swz(srow)) * 16);
  const unsigned short *aptr[CA], *wptr[CW];
  auto setup = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; i++) {
      const int am = min(m0 + srow + i * (NT / SL), g.M - 1);
      if (g.conv_C) {   // top-left pixel of the 3x3 patch in the halo buffer
        const int hw = g.conv_H * g.conv_W, b = am / hw, r = am - b * hw, y = r / g.conv_W, x = r - y * g.conv_W;
        const size_t pix = ((size_t)b * (g.conv_H + 2) + y) * (g.conv_W + 2) + x;
        aptr[i] = g.a_packed ? g.A + ((size_t)(sslot >> 2) * g.a_rows + pix) * 32 + (sslot & 3) * 8 : g.A + pix * g.conv_C + sslot * 8;
      } else if (g.a_packed) {   // slot sslot of a 64 KS-byte stage row = slice sslot / 4, 16-byte chunk sslot % 4
        aptr[i] = g.A + ((size_t)(sslot >> 2) * g.a_rows + am) * 32 + (sslot & 3) * 8;
      } else {
        aptr[i] = g.A + (size_t)am * g.K + sslot * 8;
      }
    }
#pragma unroll
    for (int i = 0; i < CW; i++) {
      const size_t wr = (size_t)min(n0 + srow + i * (NT / SL), g.N - 1);
      // packed W: slot sslot of a 64 KS-byte stage row = slice sslot / 4, 16-byte chunk sslot % 4
      wptr[i] = g.w_packed ? g.W + ((size_t)(sslot >> 2) * g.N + wr) * 32 + (sslot & 3) * 8 : g.W + wr * g.K + sslot * 8;
    }
  };
  setup();
  const bool a_on = CA * NT == TM * SL || srow < TM, w_on = CW * NT == TN * SL || srow < TN;   // tiles smaller than a pass
  u32x4 ra[NS][3][CA], rw[NS][3][CW];
  auto issue = [&](int k0, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    if ((S3_ABLATE & 1) && k0 > 0) return;
    size_t ka = g.a_packed ? (size_t)k0 * g.a_rows : (size_t)k0;
    if (g.conv_C) {   // wave-uniform: the tap this K stage belongs to (conv_C % (32 KS) == 0: a stage never straddles a tap)
      const int tap = k0 / g.conv_C, c0 = k0 - tap * g.conv_C, ky = tap / 3, kx = tap - 3 * ky;
      ka = g.a_packed ? ((size_t)(c0 >> 5) * g.a_rows + ky * (g.conv_W + 2) + kx) * 32 : (size_t)((ky * (g.conv_W + 2) + kx) * g.conv_C + c0);
    }
    const size_t kw = g.w_packed ? (size_t)k0 * g.N : (size_t)k0;   // packed: slice k0 / 32 starts (k0 / 32) N 32 elements in
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int i = 0; i < CA; i++) ra[S][p][i] = *reinterpret_cast<const u32x4 *>(aptr[i] + p * g.a_plane + ka);
#pragma unroll
      for (int i = 0; i < CW; i++) rw[S][p][i] = *reinterpret_cast<const u32x4 *>(wptr[i] + p * g.w_plane + kw);
    }
  };
  auto store = [&](int buf, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    if ((S3_ABLATE & 2) && buf > 0) return;
    unsigned char *base = smem + buf * STAGE + sw;
#pragma unroll
    for (int p = 0; p < 3; p++) {
      if (a_on) {
#pragma unroll
        for (int i = 0; i < CA; i++) *reinterpret_cast<u32x4 *>(base + p * APLANE + i * (NT / SL) * ROWB) = ra[S][p][i];
      }
      if (w_on) {
#pragma unroll
        for (int i = 0; i < CW; i++) *reinterpret_cast<u32x4 *>(base + 3 * APLANE + p * WPLANE + i * (NT / SL) * ROWB) = rw[S][p][i];
      }
    }
  };
  f32x4 acc[BM][BN];
  const int l16 = lane & 15, lq = lane >> 4;
  // fragment address of (row base + l16, slot 4 slice + lq): block bases are multiples of 16 rows, so the swizzle term is the
  // lane's own
  int foff[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ks++) foff[ks] = l16 * ROWB + (((4 * ks + lq) ^ swz(l16)) * 16);
  const int abase = wm * ROWB, bbase = 3 * APLANE + wn * ROWB;
  bf16x8 wf[3][BN], af[2][3];
  bool frag_reads = true;   // S3_ABLATE bit 3: switched off after the tile prologue
  auto read_w = [&](int buf, int j, int ks) __attribute__((always_inline)) {
    if ((S3_ABLATE & 8) && !frag_reads) return;
    const unsigned char *base = smem + buf * STAGE + bbase + j * 16 * ROWB + foff[ks];
#pragma unroll
    for (int p = 0; p < 3; p++) wf[p][j] = *reinterpret_cast<const bf16x8 *>(base + p * WPLANE);
  };
  auto read_a = [&](int buf, int i, int set, int ks) __attribute__((always_inline)) {
    if ((S3_ABLATE & 8) && !frag_reads) return;
    const unsigned char *base = smem + buf * STAGE + abase + i * 16 * ROWB + foff[ks];
#pragma unroll
    for (int p = 0; p < 3; p++) af[set][p] = *reinterpret_cast<const bf16x8 *>(base + p * APLANE);
  };
  // one 16 x 16 block over a 32-k slice: the six terms, smallest first -- THE order of the arithmetic (see the header)
  auto block = [&](int i, int j, int set) __attribute__((always_inline)) {
    f32x4 c = acc[i][j];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[set][2], c, 0, 0, 0);   // a3 w1
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][j], af[set][0], c, 0, 0, 0);   // a1 w3
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], af[set][1], c, 0, 0, 0);   // a2 w2
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[set][1], c, 0, 0, 0);   // a2 w1
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], af[set][0], c, 0, 0, 0);   // a1 w2
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[set][0], c, 0, 0, 0);   // a1 w1
    acc[i][j] = c;
  };
  constexpr int BKS = 32 * KS;
  const int nk = g.K / BKS;
  // start of a tile: its first stage is already in flight (issued here for the first tile, before the previous tile's epilogue
  // for the following ones)
  auto tile_prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < BM; i++)
#pragma unroll
      for (int j = 0; j < BN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    store(0, IntC<0>{});
    if constexpr (NS == 1) {
      if (nk > 1) issue(BKS, IntC<0>{});
    } else {
      // stages 1 .. NS into sets 1 .. NS-1, 0 (stage j lives in set j % NS; set 0 was just stored)
      s3_for<NS>([&](auto u_c) __attribute__((always_inline)) {
        constexpr int J = decltype(u_c)::value + 1;
        if (J < nk) issue(J * BKS, IntC<J % NS>{});
      });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // prologue of the rotating schedule: Whi and A row 0 of slice 0
#pragma unroll
    for (int j = BN - HB; j < BN; j++) read_w(0, j, 0);
    read_a(0, 0, 0, 0);
    if (S3_ABLATE & 8) {   // every fragment register defined once, then no more reads
#pragma unroll
      for (int j = 0; j < BN; j++) read_w(0, j, 0);
      read_a(0, 0, 1, 0);
      frag_reads = false;
    }
  };
  issue(0, IntC<0>{});
  // MFMA part of ONE 32-k slice (index KSI of its stage).  The slice's first fragments (Whi, A row 0) were read during the
  // previous slice's tail; its own tail reads the next slice's: slice KSI + 1 of the same stage, or -- behind the stage's one
  // barrier -- slice 0 of the other stage (`more`: a next stage exists).
  auto compute = [&](int cur, auto ks_c, bool more) __attribute__((always_inline)) {
    constexpr int KSI = decltype(ks_c)::value;
    constexpr bool LAST = KSI == KS - 1;
    const int nxt = cur ^ 1;
    if constexpr (BN >= 2) {
#pragma unroll
      for (int j = 0; j < BN - HB; j++) read_w(cur, j, KSI);       // Wlo: needed after the Whi blocks of row 0
    }
#pragma unroll
    for (int i = 0; i < BM; i++) {
      const int set = i & 1;
      if (i + 1 < BM) read_a(cur, i + 1, set ^ 1, KSI);
      // rows 0 and BM-1 run their Whi blocks first
#pragma unroll
      for (int j = BN - HB; j < BN; j++) block(i, j, set);
      if (i == BM - 1) {
        if constexpr (LAST) {
          if constexpr (NS == 1) {   // one LDS write / global load / fragment read per MFMA instead of clumps (measured +1.5 % end to end)
            constexpr int NMF1 = 6 * (BM * BN - (BN - HB)), NDW = 3 * (CA + CW), NDR1 = 3 * (BN - HB) + 3 * (BM - 1);
#pragma unroll
            for (int q = 0; q < NMF1; q++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              if (q < NDW) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
              else if (q < 2 * NDW) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
              if (q % 4 == 0 && q / 4 < NDR1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          // every LDS read of stage `cur` has been issued; the writes into `nxt` were issued at the top of the stage
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (!(S3_ABLATE & 4)) __builtin_amdgcn_s_barrier();
          if (more) {
#pragma unroll
            for (int j = BN - HB; j < BN; j++) read_w(nxt, j, 0);   // Whi of the next slice (its registers are free now) ...
            read_a(nxt, 0, BM == 1 ? 0 : (BM & 1), 0);              // ... and A row 0, into the set row BM-1 does not use
          }
        } else {
#pragma unroll
          for (int j = BN - HB; j < BN; j++) read_w(cur, j, KSI + 1);
          read_a(cur, 0, BM == 1 ? 0 : (BM & 1), KSI + 1);
        }
      }
      if constexpr (BN >= 2) {
#pragma unroll
        for (int j = 0; j < BN - HB; j++) block(i, j, set);
      }
    }
  };
  static_assert((BM == 1 && BN == 1) || (BM % 2) == 0, "A row 0 of the next slice must land in set 0, free at that point");
  auto tile_mainloop = [&]() __attribute__((always_inline)) {
    auto stage_mfma = [&](int cur, bool more) __attribute__((always_inline)) {
      s3_for<KS>([&](auto ks_c) __attribute__((always_inline)) { compute(cur, ks_c, more); });
    };
    if constexpr (NS == 1) {
      auto stage = [&](int kt, auto store_c, auto issue_c) __attribute__((always_inline)) {
        constexpr bool ST = decltype(store_c)::value, IS = decltype(issue_c)::value;
        if constexpr (ST) store((kt & 1) ^ 1, IntC<0>{});
        if constexpr (IS) issue((kt + 2) * BKS, IntC<0>{});
        stage_mfma(kt & 1, ST);
      };
      using T = std::true_type;
      using F = std::false_type;
      int kt = 0;
      for (; kt + 2 < nk; ++kt) stage(kt, T{}, T{});
      if (kt + 1 < nk) {
        stage(kt, T{}, F{});
        ++kt;
      }
      stage(kt, F{}, F{});
    } else {
      for (int base = 0; base < nk; base += NS) {
        s3_for<NS>([&](auto u_c) __attribute__((always_inline)) {
          constexpr int U = decltype(u_c)::value;
          const int kt = base + U;              // base % NS == 0: stage kt + 1 lives in set (U + 1) % NS
          if (kt < nk) {                        // uniform
            const bool more = kt + 1 < nk;
            if (more) {
              store((kt & 1) ^ 1, IntC<(U + 1) % NS>{});
              if (kt + 1 + NS < nk) issue((kt + 1 + NS) * BKS, IntC<(U + 1) % NS>{});
            }
            stage_mfma(kt & 1, more);
          }
        });
      }
    }
  };

  auto tile_epilogue = [&](int em0, int en0) __attribute__((always_inline)) {
    // the wave's LDS slice: its whole 16 BM x 16 BN tile with the padded row stride (the staging buffers are free after the last barrier)
    s3_tile_epilogue<BM, BN, 1>(g, acc, reinterpret_cast<float *>(smem) + wave * (16 * BM * (16 * BN + 4)), em0, en0, wm, wn, lane);
  };

#ifdef S3_STAMPS
  unsigned long long ph_pro = 0, ph_main = 0, ph_epi = 0, ph_tiles = 0;
#endif
  for (;;) {
#ifdef S3_STAMPS
    const unsigned long long ph0 = __builtin_amdgcn_s_memtime();
#endif
    tile_prologue();
#ifdef S3_STAMPS
    const unsigned long long ph1 = __builtin_amdgcn_s_memtime();
#endif
    tile_mainloop();
#ifdef S3_STAMPS
    const unsigned long long ph2 = __builtin_amdgcn_s_memtime();
#endif
    const int em0 = m0, en0 = n0;   // the tile being finished
    bool more_tiles = false;
    if constexpr (PERSIST) {
      tile += gridDim.x;
      more_tiles = tile < nwg;
      if (more_tiles) {   // the next tile's first stage flies during the epilogue below
        locate(tile);
        setup();
        issue(0, IntC<0>{});
      }
    }
    tile_epilogue(em0, en0);
#ifdef S3_STAMPS
    {
      const unsigned long long ph3 = __builtin_amdgcn_s_memtime();
      ph_pro += ph1 - ph0;
      ph_main += ph2 - ph1;
      ph_epi += ph3 - ph2;
      ph_tiles++;
    }
    if (!more_tiles && threadIdx.x == 0 && blockIdx.x < 1024) {
      s3_wg[blockIdx.x][0] = ph_pro;
      s3_wg[blockIdx.x][1] = ph_main;
      s3_wg[blockIdx.x][2] = ph_epi;
      s3_wg[blockIdx.x][3] = ph_tiles;
      s3_wg[blockIdx.x][4] = st_rt0;
      s3_wg[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
    }
    if (!more_tiles) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left the wave
      if (threadIdx.x == 0) {
        atomicMin(&s3_stamps[0], st_rt0);
        atomicMax(&s3_stamps[1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        if (blockIdx.x == 0) {
          s3_stamps[2] = st_c0;
          s3_stamps[3] = __builtin_amdgcn_s_memtime();
        }
      }
    }
#endif
    if (!more_tiles) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done with its epilogue slice before the staging buffers are refilled
  }
}


// ---- The same tile arithmetic with LDS-DMA staging (global_load_lds_dwordx4: HBM/L2 -> LDS without passing the register file).
// Round 3, from K-loop ablations of the kernel above (tools/micro/split3_phases.hip -DS3_ABLATE, profiles/round3_split3_phases.txt):
// with the 128x256 tile the VGPR -> LDS write pass (72 ds_write_b128 per slice and workgroup, ~80 B/clk/CU through the store
// path) cost 16 % of the launch and the register-staged loads another 10 %; MFMAs + fragment reads + the barrier alone run at
// the chip's power ceiling.  Here a K slice of the next stage is requested straight into the other LDS stage right after the
// barrier that frees it and awaited (vmcnt(0)) just before the next barrier: no staging registers (-36 VGPRs), no ds_write.
//  * A 1 KiB DMA piece = 16 rows x 64 B of one plane, lane l -> LDS byte 16 l of the piece (row l >> 2, physical slot l & 3);
//    the XOR swizzle of the fragment reads is applied on the SOURCE side: lane l fetches logical slot (l & 3) ^ swz(row).
//  * wave w moves pieces w, w + 8, ... of each plane (A: TM / 16 pieces, W: TN / 16), i.e. (TM + TN) / 128 pieces per plane.
//  * global addresses = wave-uniform 64-bit base (tile row 0, plane, K offset: scalar registers) + a per-lane 32-bit byte
//    offset fixed for the whole tile (row clamp, conv halo geometry and the slot permutation live in it).
//  * persistent walk: the next tile's first slice is requested into stage 0 BEFORE the epilogue, which therefore works in
//    the stage-1 region only, half of a wave's tile at a time (s3_tile_epilogue<.., 2>).
// The MFMA order per output element is the kernel's above: results are bitwise identical.
typedef __attribute__((address_space(3))) void s3_lds_void;
typedef __attribute__((address_space(1))) const void s3_glb_void;

template <int WAVES_M, int WAVES_N, int BM, int BN, bool PERSIST, bool WS1 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 1)
void gemm_split3_dma_kernel(S3Args g) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW, TM = 16 * BM * WAVES_M, TN = 16 * BN * WAVES_N;
  constexpr int ROWB = 64;
  // 1 KiB pieces per plane: of the tile / of one wave.  When NW does not divide a count (the 128x192 tile: 12 W pieces, 8 waves) the last
  // requests wrap around to pieces 0 .. -- the same bytes to the same LDS address twice, harmless -- so every wave issues the same count
  constexpr int PA = TM / 16, PW = TN / 16, CA = (PA + NW - 1) / NW, CW = (PW + NW - 1) / NW;
  constexpr int APLANE = TM * ROWB, WPLANE = TN * ROWB, STAGE = 3 * (APLANE + WPLANE);
  // WS1 (the 256x256 tile: two whole stages would be 192 KB): A double-buffered, W SINGLE-buffered -- a wave holds its W fragments of a
  // slice in registers for the whole slice, so once every wave has read them (a second barrier, behind row 0's first MFMAs) the W region
  // takes slice kt + 1 while slice kt is still being multiplied.  LDS image: A stage 0 | A stage 1 | W.
  constexpr int ASTAGE = WS1 ? 3 * APLANE : STAGE, WOFF = WS1 ? 6 * APLANE : 3 * APLANE, WSTAGE = WS1 ? 0 : STAGE;
  constexpr int LDS_TOTAL = WS1 ? 6 * APLANE + 3 * WPLANE : 2 * STAGE;
  static_assert(!(WS1 && PERSIST), "the single W buffer has no room for the next tile's first slice during the epilogue");
  constexpr int HB = BN / 2;
  static_assert(BN >= 2 && (BM % 2) == 0, "rotating fragment schedule");
  static_assert(NW * (16 * BM / 2) * (16 * BN + 4) * 4 <= (WS1 ? LDS_TOTAL : STAGE), "the two-pass epilogue must fit one staging buffer (WS1: the whole LDS)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef S3_STAMPS
  const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_c0 = __builtin_amdgcn_s_memtime();
#endif

  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN, nwg = tiles_m * tiles_n;
  int m0, n0;
  auto locate = [&](int t) __attribute__((always_inline)) {   // XCD-aware bijective remap + grouped walk, as the kernel above
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, within = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    const int per_group = 8 * tiles_n, group = t / per_group, first_m = group * 8, gsz = min(tiles_m - first_m, 8),
              in_group = t - group * per_group;
    m0 = (first_m + in_group % gsz) * TM;
    n0 = (in_group / gsz) * TN;
    if (S3_ABLATE & 16) m0 = (S3_ABLATE & 32) ? m0 : 0, n0 = 0;   // diagnostic: bit 4 = every tile reads the operands of tile (0, 0); bits 4+5 = of tile (m, 0)
  };
  int tile = blockIdx.x;
  locate(tile);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave / WAVES_N) * (16 * BM), wn = (wave % WAVES_N) * (16 * BN);
  auto swz = [](int row) __attribute__((always_inline)) {
    const int gq = (row >> 2) & 3;
    return (((gq ^ (gq >> 1)) & 1) << 1) | (gq >> 1);
  };
  // DMA source map of this lane: row (lane >> 2) of a piece, logical slot = physical slot ^ swz(row) (pieces start on multiples of 16 rows)
  const int prow = lane >> 2, pslot = (lane & 3) ^ swz(prow);
  unsigned voffA[CA], voffW[CW];          // byte offsets from the tile's uniform bases
  const unsigned short *baseA, *baseW;    // uniform: plane 0, K offset 0, tile row 0
  auto conv_off = [&](int am) __attribute__((always_inline)) {   // element offset of logical row am's top-left halo pixel
    const int hw = g.conv_H * g.conv_W, b = am / hw, r = am - b * hw, y = r / g.conv_W, x = r - y * g.conv_W;
    return (((size_t)b * (g.conv_H + 2) + y) * (g.conv_W + 2) + x) * (g.a_packed ? 32 : g.conv_C);
  };
  auto setup = [&]() __attribute__((always_inline)) {
    const size_t astride = g.a_packed ? 32 : g.K;   // elements between consecutive rows of a slice
    const size_t a0 = g.conv_C ? conv_off(min(m0, g.M - 1)) : (size_t)m0 * astride;
    baseA = g.A + a0;
    baseW = g.W + (size_t)n0 * (g.w_packed ? 32 : g.K);
#pragma unroll
    for (int i = 0; i < CA; i++) {
      const int am = min(m0 + ((wave + NW * i) % PA) * 16 + prow, g.M - 1);
      const size_t off = g.conv_C ? conv_off(am) : (size_t)am * astride;
      voffA[i] = (unsigned)((off - a0) * 2 + pslot * 16);
      if ((S3_ABLATE & 192) == 64) voffA[i] = (unsigned)(((wave + NW * i) % PA) * 1024 + lane * 16);   // diagnostic: bit 6 = operands read as if stored tile-packed (a piece = 1 KiB contiguous); bits 6+7 = W only
    }
#pragma unroll
    for (int i = 0; i < CW; i++) {
      const int wr = min(n0 + ((wave + NW * i) % PW) * 16 + prow, g.N - 1);
      voffW[i] = (unsigned)(((size_t)(wr - n0) * (g.w_packed ? 32 : g.K)) * 2 + pslot * 16);
      if (S3_ABLATE & 64) voffW[i] = (unsigned)(((wave + NW * i) % PW) * 1024 + lane * 16);
    }
  };
  setup();
  // request K slice k0 of the current tile into LDS stage `buf`: 3 (CA + CW) instructions per wave
  auto dma = [&](int k0, int buf) __attribute__((always_inline)) {
    if ((S3_ABLATE & 1) && k0 > 0) return;
    size_t ka = g.a_packed ? (size_t)k0 * g.a_rows : (size_t)k0;
    if (g.conv_C) {   // the tap this K slice belongs to (conv_C % 32 == 0: a slice never straddles a tap)
      const int tap = k0 / g.conv_C, c0 = k0 - tap * g.conv_C, ky = tap / 3, kx = tap - 3 * ky;
      ka = g.a_packed ? ((size_t)(c0 >> 5) * g.a_rows + ky * (g.conv_W + 2) + kx) * 32 : (size_t)((ky * (g.conv_W + 2) + kx) * g.conv_C + c0);
    }
    unsigned char *stageA = smem + buf * ASTAGE, *stageW = smem + WOFF + buf * WSTAGE;
#pragma unroll
    for (int p = 0; p < 3; p++) {
      if ((S3_ABLATE & 256) && p == 2 && k0 > 0) continue;   // diagnostic: bit 8 = a third fewer operand bytes per slice (what a 256x256 tile would move per flop)
      const char *pa = reinterpret_cast<const char *>(baseA + p * g.a_plane + (((S3_ABLATE & 192) == 64) ? (size_t)(k0 / 32) * (TM * 32) : ka));
      const char *pw = reinterpret_cast<const char *>(baseW + p * g.w_plane + ((S3_ABLATE & 64) ? (size_t)(k0 / 32) * (TN * 32) : (g.w_packed ? (size_t)k0 * g.N : (size_t)k0)));
#pragma unroll
      for (int i = 0; i < CA; i++)
        __builtin_amdgcn_global_load_lds((s3_glb_void *)(pa + voffA[i]), (s3_lds_void *)(stageA + p * APLANE + ((wave + NW * i) % PA) * 1024), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < CW; i++)
        __builtin_amdgcn_global_load_lds((s3_glb_void *)(pw + voffW[i]), (s3_lds_void *)(stageW + p * WPLANE + ((wave + NW * i) % PW) * 1024), 16, 0, 0);
    }
  };
  f32x4 acc[BM][BN];
  const int l16 = lane & 15, lq = lane >> 4;
  const int foff = l16 * ROWB + ((lq ^ swz(l16)) * 16);
  const int abase = wm * ROWB, bbase = WOFF + wn * ROWB;
  bf16x8 wf[3][BN], af[2][3];
  auto read_w = [&](int buf, int j) __attribute__((always_inline)) {
    const unsigned char *base = smem + buf * WSTAGE + bbase + j * 16 * ROWB + foff;
#pragma unroll
    for (int p = 0; p < 3; p++) wf[p][j] = *reinterpret_cast<const bf16x8 *>(base + p * WPLANE);
  };
  auto read_a = [&](int buf, int i, int set) __attribute__((always_inline)) {
    const unsigned char *base = smem + buf * ASTAGE + abase + i * 16 * ROWB + foff;
#pragma unroll
    for (int p = 0; p < 3; p++) af[set][p] = *reinterpret_cast<const bf16x8 *>(base + p * APLANE);
  };
  // one 16 x 16 block over a 32-k slice: the six terms, smallest first -- THE order of the arithmetic (header of this file)
  auto block = [&](int i, int j, int set) __attribute__((always_inline)) {
    f32x4 c = acc[i][j];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[set][2], c, 0, 0, 0);   // a3 w1
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[2][j], af[set][0], c, 0, 0, 0);   // a1 w3
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], af[set][1], c, 0, 0, 0);   // a2 w2
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[set][1], c, 0, 0, 0);   // a2 w1
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][j], af[set][0], c, 0, 0, 0);   // a1 w2
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][j], af[set][0], c, 0, 0, 0);   // a1 w1
    acc[i][j] = c;
  };
  const int nk = g.K / 32;
  // start of a tile: its first slice has been requested into stage 0 (here for the first tile, before the previous tile's epilogue
  // for the following ones)
  auto tile_prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < BM; i++)
#pragma unroll
      for (int j = 0; j < BN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of slice 0 have landed (and its epilogue traffic is done)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = BN - HB; j < BN; j++) read_w(0, j);
    read_a(0, 0, 0);
  };
  dma(0, 0);
  if ((S3_DMA_VARIANT & 2) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  // one K slice: [request slice kt + 1 into the other stage] MFMAs in the rotating order of the kernel above; the ONE barrier sits
  // 6 BN / 2 MFMAs before the slice's end, behind this wave's vmcnt(0) (its pieces of slice kt + 1 are in LDS) and lgkmcnt(0) (its
  // reads of the current stage are done), and the MFMAs left cover the first fragment reads of the next stage
  auto slice = [&](int kt, bool more) __attribute__((always_inline)) {
    const int cur = kt & 1, nxt = cur ^ 1;
    if (!WS1 && more) dma((kt + 1) * 32, nxt);
#pragma unroll
    for (int j = 0; j < BN - HB; j++) read_w(cur, j);       // Wlo: needed after the Whi blocks of row 0
#pragma unroll
    for (int i = 0; i < BM; i++) {
      const int set = i & 1;
      if (i + 1 < BM) read_a(cur, i + 1, set ^ 1);
#pragma unroll
      for (int j = BN - HB; j < BN; j++) block(i, j, set);
      if (WS1 && i == 0) {
        // every W fragment of this slice is in registers once the Wlo reads above have landed: behind a barrier the W region is free for
        // slice kt + 1 (row 0's Whi MFMAs are in the pipe meanwhile)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) dma((kt + 1) * 32, nxt);
      }
      if (i == BM - 1) {
        constexpr int NMF1 = 6 * (BM * BN - (BN - HB)), NDMA = 3 * (CA + CW), NDR1 = 3 * (BN - HB) + 3 * (BM - 1);
        if constexpr (!(S3_DMA_VARIANT & 1)) {
#pragma unroll
          for (int q = 0; q < NMF1; q++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < NDMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            if (q % 4 == 0 && q / 4 < NDR1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (more) {
#pragma unroll
          for (int j = BN - HB; j < BN; j++) read_w(nxt, j);
          read_a(nxt, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < BN - HB; j++) block(i, j, set);
    }
  };
#ifdef S3_STAMPS
  unsigned long long ph_pro = 0, ph_main = 0, ph_epi = 0, ph_tiles = 0;
#endif
  for (;;) {
#ifdef S3_STAMPS
    const unsigned long long ph0 = __builtin_amdgcn_s_memtime();
#endif
    tile_prologue();
#ifdef S3_STAMPS
    const unsigned long long ph1 = __builtin_amdgcn_s_memtime();
#endif
    for (int kt = 0; kt + 1 < nk; ++kt) slice(kt, true);
    slice(nk - 1, false);
#ifdef S3_STAMPS
    const unsigned long long ph2 = __builtin_amdgcn_s_memtime();
#endif
    const int em0 = m0, en0 = n0;
    bool more_tiles = false;
    if constexpr (PERSIST) {
      tile += gridDim.x;
      more_tiles = tile < nwg;
      if (more_tiles) {   // every read of both stages is behind the last barrier: the next tile's first slice flies during the epilogue
        locate(tile);
        setup();
        dma(0, 0);
      }
    }
    // the epilogue's LDS slice lives in the stage-1 region (stage 0 may be receiving the next tile)
    s3_tile_epilogue<BM, BN, 2>(g, acc, reinterpret_cast<float *>(smem + (WS1 ? 0 : STAGE)) + wave * ((16 * BM / 2) * (16 * BN + 4)), em0, en0, wm, wn, lane);
#ifdef S3_STAMPS
    {
      const unsigned long long ph3 = __builtin_amdgcn_s_memtime();
      ph_pro += ph1 - ph0;
      ph_main += ph2 - ph1;
      ph_epi += ph3 - ph2;
      ph_tiles++;
    }
    if (!more_tiles && threadIdx.x == 0 && blockIdx.x < 1024) {
      s3_wg[blockIdx.x][0] = ph_pro;
      s3_wg[blockIdx.x][1] = ph_main;
      s3_wg[blockIdx.x][2] = ph_epi;
      s3_wg[blockIdx.x][3] = ph_tiles;
      s3_wg[blockIdx.x][4] = st_rt0;
      s3_wg[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (!more_tiles) break;
    // (the next tile's prologue waits for this wave's LDS traffic and meets the other waves at its barrier before stage 1 is refilled)
  }
}

// ---- Ring kernel: the small tiles of launches that cannot fill the chip (single-image requests: M = 289, 545, 256 ...).  Such a
// launch is one workgroup per CU with one wave per SIMD, so nothing hides what a wave waits for; in the register-staged kernel above a
// 64-k stage of the 32x32 tile takes ~1000 cycles against ~260 of dependent MFMAs (tools/micro/launch_latency.hip): the LDS write ->
// barrier -> fragment read -> MFMA sequence of a stage is exposed end to end, and so is most of a load.  Here
//  * K steps (KS slices of 32 k) are requested by LDS-DMA into a RING of steps, RING - 1 steps ahead of the one being multiplied
//    (a per-wave `s_waitcnt vmcnt(n)` with n = the DMA instructions of the younger steps: the oldest step has landed, the others fly);
//  * the fragments of step s + 1 are read from LDS (into the second register set) BEFORE the MFMAs of step s are issued, so per step a
//    wave pays max(MFMA chain, LDS latency) instead of their sum, with ONE barrier;
//  * the MFMAs of a step are issued term-major over the wave's blocks (independent accumulators back to back).
// Per accumulator the sequence is the one of the kernels above (k ascending, six terms smallest first): bitwise identical results.
// Pieces, swizzle and fragment addressing as in the LDS-DMA kernel.  No convolution mode, K % (32 KS) == 0 (the dispatcher checks).
template <int WAVES_M, int WAVES_N, int BM, int BN, int KS, int RING>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 1)
void gemm_split3_ring_kernel(S3Args g) {
  constexpr int NW = WAVES_M * WAVES_N, TM = 16 * BM * WAVES_M, TN = 16 * BN * WAVES_N;
  constexpr int ROWB = 64, PA = TM / 16, PW = TN / 16, PP = PA + PW, NP = 3 * PP * KS;   // 1 KiB pieces of a step
  // every wave requests the same number C of pieces per step (its vmcnt arithmetic); when NW does not divide NP the last few requests
  // wrap around to pieces 0 .. : the same bytes to the same LDS address twice, harmless
  constexpr int C = (NP + NW - 1) / NW;
  constexpr int APLANE = TM * ROWB, WPLANE = TN * ROWB, SLICE = 3 * (APLANE + WPLANE), STEP = KS * SLICE;
  static_assert(RING >= 3 && (RING - 2) * C <= 63, "vmcnt is a 6-bit counter");
  static_assert(NW * 16 * BM * (16 * BN + 4) * 4 <= RING * STEP, "the epilogue's per-wave slices fit the ring");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN, nwg = tiles_m * tiles_n;
  int m0, n0;
  {   // XCD-aware bijective remap + grouped walk, as the kernels above
    int t = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, within = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    const int per_group = 8 * tiles_n, group = t / per_group, first_m = group * 8, gsz = min(tiles_m - first_m, 8),
              in_group = t - group * per_group;
    m0 = (first_m + in_group % gsz) * TM;
    n0 = (in_group / gsz) * TN;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave / WAVES_N) * (16 * BM), wn = (wave % WAVES_N) * (16 * BN);
  auto swz = [](int row) __attribute__((always_inline)) {
    const int gq = (row >> 2) & 3;
    return (((gq ^ (gq >> 1)) & 1) << 1) | (gq >> 1);
  };
  // this wave's pieces of a step: flat piece f = wave + NW i -> (slice, plane, A piece r | W piece r - PA); lane l fetches row l >> 2,
  // logical slot (l & 3) ^ swz(row) of the piece into LDS byte 16 l
  const int prow = lane >> 2, pslot = (lane & 3) ^ swz(prow);
  const char *src[C];
  size_t kmul[C];     // wave-uniform
  unsigned dst[C];
#pragma unroll
  for (int i = 0; i < C; i++) {
    const int f = (wave + NW * i) % NP, ks = f / (3 * PP), rem = f - ks * (3 * PP), p = rem / PP, r = rem - p * PP;
    const bool isA = r < PA;
    const int rw = isA ? min(m0 + r * 16 + prow, g.M - 1) : min(n0 + (r - PA) * 16 + prow, g.N - 1);
    const unsigned short *base = isA ? g.A + p * g.a_plane : g.W + p * g.w_plane;
    const bool packed = isA ? g.a_packed != 0 : g.w_packed != 0;
    const size_t prows = isA ? (size_t)g.a_rows : (size_t)g.N;          // rows of a slice in the slice-major layout
    src[i] = reinterpret_cast<const char *>(base + (packed ? ((size_t)ks * prows + rw) * 32 : (size_t)rw * g.K + ks * 32)) + pslot * 16;
    kmul[i] = packed ? prows * (64 * KS) : (size_t)(64 * KS);   // bytes per K step
    dst[i] = (unsigned)(ks * SLICE + (isA ? p * APLANE + r * 1024 : 3 * APLANE + p * WPLANE + (r - PA) * 1024));
  }
  auto dma = [&](int step, int slot) __attribute__((always_inline)) {
    unsigned char *sbase = smem + slot * STEP;
#pragma unroll
    for (int i = 0; i < C; i++)
      __builtin_amdgcn_global_load_lds((s3_glb_void *)(src[i] + (size_t)step * kmul[i]), (s3_lds_void *)(sbase + dst[i]), 16, 0, 0);
  };
  f32x4 acc[BM][BN];
#pragma unroll
  for (int i = 0; i < BM; i++)
#pragma unroll
    for (int j = 0; j < BN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l16 = lane & 15, lq = lane >> 4;
  const int foff = l16 * ROWB + ((lq ^ swz(l16)) * 16);
  const int abase = wm * ROWB + foff, bbase = 3 * APLANE + wn * ROWB + foff;
  bf16x8 af[2][KS][BM][3], wf[2][KS][BN][3];
  auto read_frags = [&](int slot, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    const unsigned char *sbase = smem + slot * STEP;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
#pragma unroll
      for (int p = 0; p < 3; p++) {
#pragma unroll
        for (int i = 0; i < BM; i++) af[S][ks][i][p] = *reinterpret_cast<const bf16x8 *>(sbase + ks * SLICE + p * APLANE + i * 16 * ROWB + abase);
#pragma unroll
        for (int j = 0; j < BN; j++) wf[S][ks][j][p] = *reinterpret_cast<const bf16x8 *>(sbase + ks * SLICE + p * WPLANE + j * 16 * ROWB + bbase);
      }
    }
  };
  auto compute = [&](auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    constexpr int TW[6] = {0, 2, 1, 0, 1, 0}, TA[6] = {2, 0, 1, 1, 0, 0};   // (w plane, a plane) of the six terms: a3w1 a1w3 a2w2 a2w1 a1w2 a1w1
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
      for (int t = 0; t < 6; t++)
#pragma unroll
        for (int i = 0; i < BM; i++)
#pragma unroll
          for (int j = 0; j < BN; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S][ks][j][TW[t]], af[S][ks][i][TA[t]], acc[i][j], 0, 0, 0);
  };
  const int nsteps = g.K / (32 * KS);
  // prologue: steps 0 .. RING-2 requested, step 0 awaited, its fragments read
  {
    const int npro = min(RING - 1, nsteps);
    for (int s = 0; s < npro; s++) dma(s, s);
    if (npro == RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 2) * C) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, IntC<0>{});
  }
  int slot_next = 1 % RING;          // slot of step s + 1
  int slot_fill = (RING - 1) % RING; // slot step s + RING - 1 goes to (= the slot of step s - 1)
  auto iter = [&](int s, auto set_c) __attribute__((always_inline)) {
    constexpr int S = decltype(set_c)::value;
    if (s + 1 < nsteps) {
      // step s + 1 has landed: RING - 3 younger steps may still fly (fewer were requested near the end: wait for everything there)
      if (s + RING - 2 <= nsteps - 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 3) * C) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // ... for every wave's pieces; and every wave is past its reads of step s - 1 (and s)
      if (s + RING - 1 < nsteps) dma(s + RING - 1, slot_fill);
      read_frags(slot_next, IntC<S ^ 1>{});
      slot_next = slot_next + 1 == RING ? 0 : slot_next + 1;
      slot_fill = slot_fill + 1 == RING ? 0 : slot_fill + 1;
    }
    compute(set_c);
  };
  for (int s = 0; s < nsteps; s += 2) {
    iter(s, IntC<0>{});
    if (s + 1 < nsteps) iter(s + 1, IntC<1>{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // every wave's fragment reads are done: the ring becomes the epilogue's scratch
  s3_tile_epilogue<BM, BN, 1>(g, acc, reinterpret_cast<float *>(smem) + wave * (16 * BM * (16 * BN + 4)), m0, n0, wm, wn, lane);
}

// x = x1 + x2 + x3: planes [3][rows][cols] (bf16 bit patterns).  cols % 8 == 0, 16-byte aligned rows.
extern "C" int sgic_split3_f32(const float *d_x, int ld, int rows, int cols, int seg, int seg_stride, uint16_t *d_planes,
                               sgic_stream_t stream) {
  SGIC_REQUIRE(d_x && d_planes && rows > 0 && cols > 0, "null/empty");
  SGIC_REQUIRE((cols & 7) == 0 && (ld & 3) == 0 && ld >= cols, "cols % 8, ld % 4");
  SGIC_REQUIRE(((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_planes & 15) == 0, "16-byte alignment");
  SGIC_REQUIRE(seg >= 0 && (seg == 0 || seg_stride >= seg), "row segment map");
  const long total = (long)rows * (cols >> 3);
  const unsigned grid = (unsigned)min((total + 255) / 256, 256L * 16);
  split3_rows_kernel<false><<<grid, 256, 0, to_stream(stream)>>>(d_x, ld, rows, cols, seg, seg_stride, d_planes);
  return sgic::check_launch("split3_rows_kernel");
}

// The planes of a CONSTANT W operand in the slice-major layout [3][cols / 32][rows][32] (opts->w_packed of the GEMM / convolution
// entry points): the 16 rows x 64 B piece a wave requests per instruction is then 1 KiB of consecutive bytes instead of sixteen
// half cache lines 2 cols bytes apart.  Measured (tools/micro/cu_l2_bandwidth.hip, split3_phases -DS3_ABLATE=64): a CU pulls 76 GB/s
// in the strided shape against 105-120 GB/s contiguous with the chip loaded; the 128x256 tile gains 6-13 % with packed operands.
extern "C" int sgic_split3_pack_f32(const float *d_x, int ld, int rows, int cols, uint16_t *d_planes, sgic_stream_t stream) {
  SGIC_REQUIRE(d_x && d_planes && rows > 0 && cols > 0, "null/empty");
  SGIC_REQUIRE((cols & 31) == 0 && (ld & 3) == 0 && ld >= cols, "cols % 32, ld % 4");
  SGIC_REQUIRE(((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_planes & 15) == 0, "16-byte alignment");
  const long total = (long)rows * (cols >> 3);
  const unsigned grid = (unsigned)min((total + 255) / 256, 256L * 16);
  split3_rows_kernel<true><<<grid, 256, 0, to_stream(stream)>>>(d_x, ld, rows, cols, 0, 0, d_planes);
  return sgic::check_launch("split3_rows_kernel<pack>");
}

#define SGIC_SPLIT3_TILE_MODES 31

template <int WAVES_M, int WAVES_N, int BM, int BN, int NS, int KS = 1, bool PERSIST = false>
static int s3_launch(const S3Args &g, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  constexpr int NT = 64 * WAVES_M * WAVES_N, TM = 16 * BM * WAVES_M, TN = 16 * BN * WAVES_N;
  constexpr int LDS = 2 * 3 * (TM + TN) * 64 * KS;
  // the dynamic-LDS limit is a per-DEVICE attribute of the function: remembered per device (a process that launches on a
  // second GPU raises it there too); idempotent, so a race only sets the same value twice
  static bool attr_set[64] = {};
  auto kernel = gemm_split3_kernel<WAVES_M, WAVES_N, BM, BN, NS, KS, PERSIST>;
  int dev = 0;
  SGIC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    SGIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  unsigned ntiles = (unsigned)(((g.M + TM - 1) / TM) * ((g.N + TN - 1) / TN));
  const dim3 grid(PERSIST ? (ntiles < 256u ? ntiles : 256u) : ntiles);   // persistent: one resident workgroup per CU
  if (ev_start || ev_stop) {
    hipExtLaunchKernelGGL(kernel, grid, dim3(NT), LDS, st, ev_start, ev_stop, 0, g);
  } else {
    kernel<<<grid, NT, LDS, st>>>(g);
  }
  return sgic::check_launch("gemm_split3_kernel");
}

template <int WAVES_M, int WAVES_N, int BM, int BN, bool PERSIST = false, bool WS1 = false>
static int s3_launch_dma(const S3Args &g, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  constexpr int NT = 64 * WAVES_M * WAVES_N, TM = 16 * BM * WAVES_M, TN = 16 * BN * WAVES_N;
  constexpr int LDS = WS1 ? 3 * (2 * TM + TN) * 64 : 2 * 3 * (TM + TN) * 64;
  static_assert(LDS <= 160 * 1024, "LDS of a CU");
  static bool attr_set[64] = {};   // per device, as s3_launch
  auto kernel = gemm_split3_dma_kernel<WAVES_M, WAVES_N, BM, BN, PERSIST, WS1>;
  int dev = 0;
  SGIC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    SGIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  unsigned ntiles = (unsigned)(((g.M + TM - 1) / TM) * ((g.N + TN - 1) / TN));
  const dim3 grid(PERSIST ? (ntiles < 256u ? ntiles : 256u) : ntiles);
  if (ev_start || ev_stop) {
    hipExtLaunchKernelGGL(kernel, grid, dim3(NT), LDS, st, ev_start, ev_stop, 0, g);
  } else {
    kernel<<<grid, NT, LDS, st>>>(g);
  }
  return sgic::check_launch("gemm_split3_dma_kernel");
}

template <int WAVES_M, int WAVES_N, int BM, int BN, int KS, int RING>
static int s3_launch_ring(const S3Args &g, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  constexpr int NT = 64 * WAVES_M * WAVES_N, TM = 16 * BM * WAVES_M, TN = 16 * BN * WAVES_N;
  constexpr int LDS = RING * KS * 3 * (TM + TN) * 64;
  static_assert(LDS <= 160 * 1024, "LDS of a CU");
  static bool attr_set[64] = {};   // per device, as s3_launch
  auto kernel = gemm_split3_ring_kernel<WAVES_M, WAVES_N, BM, BN, KS, RING>;
  int dev = 0;
  SGIC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    SGIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const dim3 grid((unsigned)(((g.M + TM - 1) / TM) * ((g.N + TN - 1) / TN)));
  if (ev_start || ev_stop) {
    hipExtLaunchKernelGGL(kernel, grid, dim3(NT), LDS, st, ev_start, ev_stop, 0, g);
  } else {
    kernel<<<grid, NT, LDS, st>>>(g);
  }
  return sgic::check_launch("gemm_split3_ring_kernel");
}

// Which kernel a tile mode runs (round 3, tools/micro/split3_phases.hip on the model's shapes): the 128x256 / 256x128 tiles take the
// LDS-DMA kernel (a slice is ~3000 cycles: the DMA latency hides; 3-9 % shorter launches than register staging), the 128x128 tile
// keeps register staging under its old numbers (its slices are half as long and long-K shapes lose ~10 % with one slice of cover)
// and is ALSO offered with DMA staging as modes 16 / 17 for the tuner (short-K shapes gain 2-4 %).
static int s3_mode(const S3Args &g, int mode, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  // a DMA piece is addressed by per-lane 32-bit byte offsets from the tile's base: true for every tile below unless rows are absurdly long
  const bool dma_ok = (size_t)(g.a_packed ? 32 : g.K) * 2 * 256 < (1ull << 31) && (size_t)(g.w_packed ? 32 : g.K) * 2 * 256 < (1ull << 31) &&
                      (!g.conv_C || (size_t)(g.conv_W + 2) * (g.a_packed ? 32 : g.conv_C) * 2 * 1024 < (1ull << 31));
  switch (mode) {
    case 1: return dma_ok ? s3_launch_dma<2, 4, 4, 4>(g, st, e0, e1) : s3_launch<2, 4, 2, 2, 1>(g, st, e0, e1);
    case 10: return dma_ok ? s3_launch_dma<2, 4, 4, 4, true>(g, st, e0, e1) : s3_launch<2, 4, 2, 2, 1>(g, st, e0, e1);
#ifndef S3_MICRO_BIG_ONLY   // diagnostic builds (tools/micro) instantiate the two kernels above only: minutes less of compile time
    case 14: return dma_ok ? s3_launch_dma<4, 2, 4, 4>(g, st, e0, e1) : s3_launch<2, 4, 2, 2, 1>(g, st, e0, e1);               // 256x128: the 128x256 tile's blocking for N = 128
    case 15: return dma_ok ? s3_launch_dma<4, 2, 4, 4, true>(g, st, e0, e1) : s3_launch<2, 4, 2, 2, 1>(g, st, e0, e1);
    // 28 / 29: 128x192 (plain / persistent): N = 768 = 4 x 192 -- M = 8 192 is 64 x 4 = 256 tiles, ONE whole round, where 128x256 tiles
    // are 192 (a quarter of the chip idle).  LDS-DMA staging only; otherwise the 128x128 register tile
    case 28: return dma_ok ? s3_launch_dma<2, 4, 4, 3>(g, st, e0, e1) : s3_launch<2, 4, 4, 2, 1>(g, st, e0, e1);
    case 29: return dma_ok ? s3_launch_dma<2, 4, 4, 3, true>(g, st, e0, e1) : s3_launch<2, 4, 4, 2, 1>(g, st, e0, e1);
    // 30: 256x256 (W single-buffered: 144 KB): a third fewer operand bytes per flop than 128x256
    case 30: return dma_ok ? s3_launch_dma<2, 4, 8, 4, false, true>(g, st, e0, e1) : s3_launch<2, 4, 2, 2, 1>(g, st, e0, e1);
    case 16: return dma_ok ? s3_launch_dma<2, 4, 4, 2>(g, st, e0, e1) : s3_launch<2, 4, 4, 2, 1>(g, st, e0, e1);
    case 17: return dma_ok ? s3_launch_dma<2, 4, 4, 2, true>(g, st, e0, e1) : s3_launch<2, 4, 4, 2, 1, 1, true>(g, st, e0, e1);
    case 2: return s3_launch<2, 4, 4, 2, 1>(g, st, e0, e1);
    // 18-22: the ring kernel (small tiles, deep LDS-DMA prefetch) for launches that cannot fill the chip; shapes it does not take
    // (a convolution, K not a multiple of its step) run the register-staged tile of the same size
    case 18: return (!g.conv_C && g.K % 64 == 0) ? s3_launch_ring<2, 2, 1, 1, 2, 6>(g, st, e0, e1) : s3_mode(g, 4, st, e0, e1);   // 32x32
    case 19: return (!g.conv_C && g.K % 64 == 0) ? s3_launch_ring<2, 2, 2, 1, 2, 4>(g, st, e0, e1) : s3_mode(g, 4, st, e0, e1);   // 64x32
    case 20: return (!g.conv_C && g.K % 64 == 0) ? s3_launch_ring<2, 2, 1, 2, 2, 4>(g, st, e0, e1) : s3_mode(g, 4, st, e0, e1);   // 32x64
    case 21: return !g.conv_C ? s3_launch_ring<2, 2, 2, 2, 1, 6>(g, st, e0, e1) : s3_mode(g, 3, st, e0, e1);                       // 64x64, four waves
    case 22: return !g.conv_C ? s3_launch_ring<2, 4, 2, 1, 1, 6>(g, st, e0, e1) : s3_mode(g, 3, st, e0, e1);                       // 64x64, eight waves
    // 23-25: tiles cut for ONE image's token count (M = 289 -> 19 row blocks: four tiles of 80 rows, or ten of 32) so that N / TN column
    // tiles x those make <= 256 workgroups, one per CU: a launch of this size is bound by what ONE CU can pull from L2 (~55 GB/s measured,
    // whatever the depth of the ring), i.e. by (TM + TN) K bytes per workgroup with every CU busy
    case 23: return !g.conv_C ? s3_launch_ring<1, 4, 5, 1, 1, 5>(g, st, e0, e1) : s3_mode(g, 3, st, e0, e1);                       // 80x64
    case 24: return !g.conv_C ? s3_launch_ring<1, 3, 5, 1, 1, 6>(g, st, e0, e1) : s3_mode(g, 3, st, e0, e1);                       // 80x48
    case 25: return (!g.conv_C && g.K % 64 == 0) ? s3_launch_ring<2, 3, 1, 1, 2, 5>(g, st, e0, e1) : s3_mode(g, 4, st, e0, e1);   // 32x48
    case 11: return s3_launch<2, 4, 4, 2, 1, 1, true>(g, st, e0, e1);
    // the 32x32 latency tile: 64-k stages when K allows (and, for a convolution, a stage stays inside one tap)
    case 3: return s3_launch<2, 2, 2, 2, 4>(g, st, e0, e1);   // (64-k stages measured no better for the 64x64 tile)
    case 4: return (g.K % 64 == 0 && g.conv_C % 64 == 0) ? s3_launch<2, 2, 1, 1, 3, 2>(g, st, e0, e1) : s3_launch<2, 2, 1, 1, 6>(g, st, e0, e1);
    default: return s3_launch<2, 4, 2, 2, 1>(g, st, e0, e1);   // 5: 64x128, two workgroups per CU
#else
    default: return SGIC_EINVAL;
#endif
  }
}

static int s3_dispatch(const S3Args &g, const sgic_launch_opts *opts, hipStream_t st) {
  const int M = g.M, N = g.N, K = g.K;
  int mode = opts ? opts->tile_mode : 0;
  SGIC_REQUIRE(mode >= 0 && mode <= SGIC_SPLIT3_TILE_MODES, "tile_mode");
  if (!mode) {
    // the largest tile whose grid still fills the 256 CUs ~twice over; launches that cannot fill the chip take the small tiles
    const long t256 = (long)((M + 127) / 128) * ((N + 255) / 256), t128 = (long)((M + 127) / 128) * ((N + 127) / 128),
               t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    mode = t256 >= 512 ? 1 : (t128 >= 384 ? 2 : (t64 >= 256 ? 3 : 4));
  }
  const auto *evp = prof_next(opts);
  hipEvent_t e0 = evp ? evp->first : nullptr, e1 = evp ? evp->second : nullptr;
  if (mode == 31) {
    // 31: 256x256 tiles (mode 30) for the rows that fill WHOLE rounds of the 256 CUs + 128x256 tiles (mode 1) for the rest
    const long tiles_n = (N + 255) / 256, tiles_m = (M + 255) / 256;
    const long m_full = (tiles_m * tiles_n / 256) * 256 / tiles_n, m_split = m_full * 256;
    if (m_full > 0 && m_split < M && !g.conv_C) {
      S3Args g1 = g, g2 = g;
      g1.M = (int)m_split;
      g2.M = M - (int)m_split;
      g2.A += g.a_packed ? m_split * 32 : m_split * K;
      g2.m_base = (int)m_split;
      int rc = s3_mode(g1, 30, st, e0, nullptr);
      if (rc) return rc;
      return s3_mode(g2, 1, st, nullptr, e1);
    }
    mode = 30;
  }
  if ((mode >= 6 && mode <= 9) || mode == 12 || mode == 13 || mode == 26 || mode == 27) {
    // modes 6 / 7 = modes 1 / 2 for the rows that fill WHOLE rounds of the 256 CUs + the 64x128 two-per-CU tiles for the rest,
    // as two launches (the first ends on a full round, so nothing idles at the seam): a last round of big tiles that covers a
    // fraction of the chip costs a whole tile time, the small tiles finish it in about half.  One profiler record spans both.
    // modes 8 / 9: the same split with the 32x32 latency tiles (64-k stages, three workgroups per CU) for the remaining rows
    // modes 12 / 13: as 8 / 9 with the whole rounds walked by the persistent launch (10 / 11)
    // modes 26 / 27: as 8 / 12 (128x256 for the whole rounds, plain / persistent) with the ring kernel's 80x64 tiles (mode 23) for the
    // remaining rows: the remainder of M = 9 248 (1 056 rows x 1024 x 4096) takes 78 us there against 103 us with the 32x32 tiles
    const bool ring_tail = mode >= 26;
    const int cm = ring_tail ? (mode == 26 ? 8 : 12) : mode;
    const int big = cm >= 12 ? cm - 11 : (cm >= 8 ? cm - 7 : cm - 5), tail = ring_tail ? 23 : (cm >= 8 ? 4 : 5), TM = 128, TN = big == 1 ? 256 : 128;
    const int big_mode = cm >= 12 ? big + 9 : big;
    const long tiles_n = (N + TN - 1) / TN, tiles_m = (M + TM - 1) / TM;
    const long m_full = (tiles_m * tiles_n / 256) * 256 / tiles_n;   // m-tiles inside whole rounds
    const long m_split = m_full * TM;
    if (m_full > 0 && m_split < M && !g.conv_C) {
      S3Args g1 = g, g2 = g;
      g1.M = (int)m_split;
      g2.M = M - (int)m_split;
      g2.A += g.a_packed ? m_split * 32 : m_split * K;   // the planes keep their strides (a_plane, a_rows); C / R / Cp rows are addressed through m_base
      g2.m_base = (int)m_split;
      int rc = s3_mode(g1, big_mode, st, e0, nullptr);
      if (rc) return rc;
      return s3_mode(g2, tail, st, nullptr, e1);
    }
    mode = big_mode;
  }
  return s3_mode(g, mode, st, e0, e1);
}

// C[M,N] = act(A . W^T + bias) + R with both operands as bf16x3 planes (see the header of this file).
//   d_A != NULL: A (fp32, lda, row map a_seg) is split into d_Aplanes (caller's workspace, 3*M*K bf16) first;
//   d_A == NULL: d_Aplanes already holds the planes (written by sgic_split3_f32 or a producing kernel).
//   d_Cplanes != NULL: the result is written as planes [3][M][N] (the A operand of the next GEMM) instead of d_C.
// K % 32 == 0.  opts->tile_mode: 0 = heuristic, 1 = 128x256, 2 = 128x128, 3 = 64x64, 4 = 32x32 tiles (the latency kernel for
// under-filled launches: 4 waves of one 16x16 block each, a dependent chain of 6 MFMAs per 32 k), 5 = 64x128 tiles with two
// workgroups per CU (finer rounds for grids that are not a multiple of the chip), 6 / 7 = 1 / 2 for the rows that fill whole rounds
// of the chip + mode 5 for the remaining rows (two launches), 8 / 9 = the same with mode 4 for the remaining rows, 10 / 11 = 1 / 2
// as a persistent launch (256 resident workgroups walk the tiles, the next tile's first loads fly during the epilogue), 12 / 13 =
// 8 / 9 with the whole rounds walked persistently, 14 / 15 = 256x128 tiles (plain / persistent: the 3x3 convolutions with 128
// output channels); bitwise identical results.
extern "C" int sgic_gemm_split3_f32(const float *d_A, int lda, int a_seg, int a_seg_stride, uint16_t *d_Aplanes,
                                    const uint16_t *d_Wplanes, const float *d_bias, const float *d_R, int ldr, float *d_C,
                                    int ldc, uint16_t *d_Cplanes, int M, int N, int K, int act, int c_seg, int c_seg_stride,
                                    const sgic_launch_opts *opts, sgic_stream_t stream) {
  SGIC_REQUIRE(d_Aplanes && d_Wplanes && (d_C || d_Cplanes) && M > 0 && N > 0 && K > 0, "null/empty");
  SGIC_REQUIRE((K & 31) == 0, "K must be a multiple of 32");
  SGIC_REQUIRE(((uintptr_t)d_Aplanes & 15) == 0 && ((uintptr_t)d_Wplanes & 15) == 0, "planes must be 16-byte aligned");
  SGIC_REQUIRE((d_Cplanes || ldc >= N) && (!d_R || ldr >= N), "leading dimensions");
  SGIC_REQUIRE(act >= 0 && act <= ACT_LRELU, "activation");
  SGIC_REQUIRE(c_seg >= 0 && (c_seg == 0 || c_seg_stride >= c_seg), "row segment map");
  int a_packed = opts ? opts->a_packed : 0;
  if (d_A) {   // the split pass writes the slice-major layout (the row map is applied on the source side)
    SGIC_REQUIRE((lda & 3) == 0 && lda >= K && ((uintptr_t)d_A & 15) == 0 && a_seg >= 0 && (a_seg == 0 || a_seg_stride >= a_seg), "A operand");
    const long total = (long)M * (K >> 3);
    split3_rows_kernel<true><<<(unsigned)min((total + 255) / 256, 256L * 16), 256, 0, to_stream(stream)>>>(d_A, lda, M, K, a_seg, a_seg_stride, d_Aplanes);
    int rc = sgic::check_launch("split3_rows_kernel<pack>");
    if (rc) return rc;
    a_packed = 1;
  }
  SGIC_REQUIRE(!d_Cplanes || (N & 31) == 0, "planes output needs N % 32 == 0");
  const int vec = (N % 4 == 0) && (!d_bias || ((uintptr_t)d_bias & 15) == 0) && (!d_R || ((ldr % 4 == 0) && ((uintptr_t)d_R & 15) == 0)) &&
                  (d_Cplanes ? ((uintptr_t)d_Cplanes & 7) == 0 : ((ldc % 4 == 0) && ((uintptr_t)d_C & 15) == 0));
  SGIC_REQUIRE(!d_Cplanes || (vec && c_seg == 0), "planes output needs float4-addressable operands and no row map");
  S3Args g{d_Aplanes, d_Wplanes, d_bias, d_R, d_C, M, N, K, ldr, ldc, act, c_seg, c_seg_stride, (long)M * K, (long)N * K, d_Cplanes, (long)M * N, vec, 0, 0, 0, 0,
           opts ? opts->w_packed : 0, a_packed, M, M};
  return s3_dispatch(g, opts, to_stream(stream));
}

// 3x3 stride-1 pad-1 convolution as an implicit split GEMM (taming ResnetBlock / Upsample / conv_in, model.py:38-137,436-537;
// the call sites of sgic_conv3x3_f32): d_in_planes = bf16x3 planes [3][halo_rows][Cin] of the zero-halo NHWC input
// (halo_rows = B (H+2) (W+2)), d_Wplanes = planes [3][Cout][9 Cin] of the (ky,kx,cin)-ordered weight.  Cin % 32 == 0.
extern "C" int sgic_conv3x3_split3_f32(const uint16_t *d_in_planes, const uint16_t *d_Wplanes, const float *d_bias,
                                       const float *d_R, int ldr, float *d_out, int ldc, int B, int H, int W, int Cin, int Cout,
                                       int act, const sgic_launch_opts *opts, sgic_stream_t stream) {
  const long Ml = (long)B * H * W, halo_rows = (long)B * (H + 2) * (W + 2);
  SGIC_REQUIRE(d_in_planes && d_Wplanes && d_out && B > 0 && H > 0 && W > 0 && Cout > 0, "null/empty");
  SGIC_REQUIRE(Ml < (1l << 31), "too many pixels");
  SGIC_REQUIRE(Cin > 0 && Cin % 32 == 0, "implicit-GEMM conv needs Cin % 32 == 0");
  SGIC_REQUIRE(((uintptr_t)d_in_planes & 15) == 0 && ((uintptr_t)d_Wplanes & 15) == 0, "planes must be 16-byte aligned");
  SGIC_REQUIRE(ldc >= Cout && (!d_R || ldr >= Cout) && act >= 0 && act <= ACT_LRELU, "leading dimensions / activation");
  const int M = (int)Ml, N = Cout, K = 9 * Cin;
  const int vec = (N % 4 == 0) && (!d_bias || ((uintptr_t)d_bias & 15) == 0) && (!d_R || ((ldr % 4 == 0) && ((uintptr_t)d_R & 15) == 0)) &&
                  (ldc % 4 == 0) && ((uintptr_t)d_out & 15) == 0;
  S3Args g{d_in_planes, d_Wplanes, d_bias, d_R, d_out, M, N, K, ldr, ldc, act, 0, 0, halo_rows * Cin, (long)N * K, nullptr, 0, vec, Cin, H, W, 0,
           opts ? opts->w_packed : 0, opts ? opts->a_packed : 0, (int)halo_rows, 0};
  return s3_dispatch(g, opts, to_stream(stream));
}
