// Experiment: fp32-accurate GEMM on the bf16 matrix pipe by a 3-way operand split ("bf16x3").
//   a = a1 + a2 + a3 (each a bf16, 8 significant bits; the three carry all 24 bits of the fp32 value), same for w;
//   a.w ~= a1w1 + (a1w2 + a2w1) + (a2w2 + a1w3 + a3w1)         -- the dropped terms are <= 2^-26 of the product
// i.e. 6 bf16 MFMAs (v_mfma_f32_32x32x16_bf16, fp32 accumulate) in place of 8 fp32 MFMAs per 16 k, each 16x the rate.
// Prints (a) the error of this scheme and of a plain fp32 fmaf chain against an fp64 reference and (b) launch times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o /tmp/split3 tools/micro/split3_gemm.hip && /tmp/split3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef SPLIT_TERMS
#define SPLIT_TERMS 6
#endif
#ifndef SEP_ACC
#define SEP_ACC 0
#endif

#define S_BK 32
#define S_ROWB 80                 // bytes per LDS row: 64 data + 16 pad (ds_read_b128 of 16 consecutive rows: 16 distinct slots)
#define S_PLANE (128 * S_ROWB)    // one plane of a 128-row operand tile
#define S_STAGE (6 * S_PLANE)     // A1 A2 A3 W1 W2 W3

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
  bf16x2 r = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &p1, unsigned &p2, unsigned &p3) {
  p1 = pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
  p2 = pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
  p3 = pk_bf16(s0, s1);
}

// W[N][K] fp32 -> three bf16 planes [N][K]
__global__ void split_planes_kernel(const float *__restrict__ W, unsigned short *__restrict__ P1, unsigned short *__restrict__ P2,
                                    unsigned short *__restrict__ P3, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    unsigned a, b, c;
    split_pair(W[2 * i], W[2 * i + 1], a, b, c);
    reinterpret_cast<unsigned *>(P1)[i] = a;
    reinterpret_cast<unsigned *>(P2)[i] = b;
    reinterpret_cast<unsigned *>(P3)[i] = c;
  }
}

struct SArgs {
  const float *A;
  const unsigned short *W1, *W2, *W3;
  float *C;
  int M, N, K, lda, ldw, ldc;
  int diag;   // 0 normal, 1 no MFMA phase, 2 no staging after the prologue
  long long *stamps;   // per block 8 values, or null
};

#define LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)

// SPEC = false: 256 threads, every wave stages and computes.  SPEC = true: 512 threads, waves 0-3 compute (64x64 each),
// waves 4-7 only stage (global loads, the 3-way split of A, LDS writes) one K slice ahead.
template <bool SPEC>
__global__ __launch_bounds__(SPEC ? 512 : 256, 1) void gemm_split3_kernel(SArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles_m = (g.M + 127) / 128, tiles_n = (g.N + 127) / 128, nwg = tiles_m * tiles_n;
  int m0, n0;
  {
    int t = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, within = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    const int per_group = 8 * tiles_n, group = t / per_group, first_m = group * 8, gsz = min(tiles_m - first_m, 8), in_group = t - group * per_group;
    m0 = (first_m + in_group % gsz) * 128;
    n0 = (in_group / gsz) * 128;
  }
  const bool producer = SPEC && threadIdx.x >= 256, consumer = !SPEC || threadIdx.x < 256;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // staging A: 128 rows x 8 float4 chunks: thread -> (row tid>>3 + 32 i, chunk tid&7)
  const int arow = tid >> 3, achunk = tid & 7;
  const float *aptr[4];
#pragma unroll
  for (int i = 0; i < 4; i++) aptr[i] = g.A + (size_t)min(m0 + arow + 32 * i, g.M - 1) * g.lda + achunk * 4;
  // staging W planes: 128 rows x 4 chunks of 16 B: thread -> (row tid>>2 + 64 i, chunk tid&3)
  const int wrow = tid >> 2, wchunk = tid & 3;
  size_t woff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) woff[i] = (size_t)min(n0 + wrow + 64 * i, g.N - 1) * g.ldw + wchunk * 8;
  f32x4 ra[4];
  u32x4 rw[3][2];
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; i++) ra[i] = *reinterpret_cast<const f32x4 *>(aptr[i] + k0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      rw[0][i] = *reinterpret_cast<const u32x4 *>(g.W1 + woff[i] + k0);
      rw[1][i] = *reinterpret_cast<const u32x4 *>(g.W2 + woff[i] + k0);
      rw[2][i] = *reinterpret_cast<const u32x4 *>(g.W3 + woff[i] + k0);
    }
  };
  auto store = [&](int buf) {
    unsigned char *base = smem + buf * S_STAGE;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      unsigned p1a, p2a, p3a, p1b, p2b, p3b;
      split_pair(ra[i][0], ra[i][1], p1a, p2a, p3a);
      split_pair(ra[i][2], ra[i][3], p1b, p2b, p3b);
      unsigned char *q = base + (arow + 32 * i) * S_ROWB + achunk * 8;
      *reinterpret_cast<u32x2 *>(q) = u32x2{p1a, p1b};
      *reinterpret_cast<u32x2 *>(q + S_PLANE) = u32x2{p2a, p2b};
      *reinterpret_cast<u32x2 *>(q + 2 * S_PLANE) = u32x2{p3a, p3b};
    }
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int i = 0; i < 2; i++) *reinterpret_cast<u32x4 *>(base + (3 + p) * S_PLANE + (wrow + 64 * i) * S_ROWB + wchunk * 16) = rw[p][i];
  };
  f32x16 acc[2][2];
#if SEP_ACC
  f32x16 acl[2][2];
#endif
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        acc[i][j][e] = 0.f;
#if SEP_ACC
        acl[i][j][e] = 0.f;
#endif
      }
  const int lrow = lane & 31, lhalf = lane >> 5;
  const int aoff = (wm + lrow) * S_ROWB + lhalf * 16, boff = 3 * S_PLANE + (wn + lrow) * S_ROWB + lhalf * 16;
  auto compute = [&](int buf) {
    const unsigned char *base = smem + buf * S_STAGE;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      bf16x8 fa[3][2], fb[3][2];
#pragma unroll
      for (int p = 0; p < 3; p++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
          fa[p][i] = *reinterpret_cast<const bf16x8 *>(base + p * S_PLANE + aoff + i * 32 * S_ROWB + s * 32);
          fb[p][i] = *reinterpret_cast<const bf16x8 *>(base + p * S_PLANE + boff + i * 32 * S_ROWB + s * 32);
        }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
#if SEP_ACC
#define LO acl
#else
#define LO acc
#endif
          if (SPLIT_TERMS >= 6) {
            LO[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2][i], fb[0][j], LO[i][j], 0, 0, 0);
            LO[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[2][j], LO[i][j], 0, 0, 0);
            LO[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[1][j], LO[i][j], 0, 0, 0);
          }
          if (SPLIT_TERMS >= 3) {
            LO[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][i], fb[0][j], LO[i][j], 0, 0, 0);
            LO[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[1][j], LO[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][i], fb[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };
  const int nk = g.K / S_BK;
#ifdef PRIO
  if (SPEC) { if (producer) __builtin_amdgcn_s_setprio(PRIO); else __builtin_amdgcn_s_setprio(0); }
#endif
  if (!SPEC || producer) {
    issue(0);
    store(0);
    if (nk > 1) issue(S_BK);
  }
  LDS_BARRIER();
  long long t_work = 0, t_bar = 0, t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const long long t0 = __builtin_amdgcn_s_memtime();
    if ((!SPEC || producer) && kt + 1 < nk && g.diag != 2) {
      store(cur ^ 1);
      if (kt + 2 < nk) issue((kt + 2) * S_BK);
    }
    if (consumer && g.diag != 1) compute(cur);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const long long t2 = __builtin_amdgcn_s_memtime();
    t_work += t1 - t0;
    t_bar += t2 - t1;
  }
  if (g.stamps && (threadIdx.x == 0 || threadIdx.x == 256)) {
    long long *q = g.stamps + (size_t)blockIdx.x * 8 + (threadIdx.x ? 4 : 0);
    q[0] = t_work;
    q[1] = t_bar;
    q[2] = __builtin_amdgcn_s_memtime() - t_begin;
    q[3] = __builtin_amdgcn_s_memrealtime() - r_begin;
  }
  if (!consumer) return;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n = n0 + wn + j * 32 + lrow;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        float v = acc[i][j][e];
#if SEP_ACC
        v += acl[i][j][e];
#endif
        if (m < g.M && n < g.N) g.C[(size_t)m * g.ldc + n] = v;
      }
    }
}


// ---- variant B: both operands pre-split into bf16 planes; 128 x 256 tile, 8 waves (2 x 4) of 64 x 64, all waves stage and compute;
// LDS rows are 64 B (32 k) unpadded, the 16-byte slot of a row is XOR-swizzled with (row >> 2) & 3 (conflict-free ds_read_b128
// lane groups and ds_write_b128), two stages of 72 KB.
struct BArgs {
  const unsigned short *A1, *A2, *A3, *W1, *W2, *W3;
  float *C;
  int M, N, K, lda, ldw, ldc;
  long long *stamps;
};
#define B_TM 128
#define B_TN 256
#define B_APLANE (B_TM * 64)
#define B_WPLANE (B_TN * 64)
#define B_STAGE (3 * B_APLANE + 3 * B_WPLANE)

__global__ __launch_bounds__(512, 1) void gemm_split3b_kernel(BArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tiles_m = (g.M + B_TM - 1) / B_TM, tiles_n = (g.N + B_TN - 1) / B_TN, nwg = tiles_m * tiles_n;
  int m0, n0;
  {
    int t = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = t & 7, within = t >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
    const int per_group = 8 * tiles_n, group = t / per_group, first_m = group * 8, gsz = min(tiles_m - first_m, 8), in_group = t - group * per_group;
    m0 = (first_m + in_group % gsz) * B_TM;
    n0 = (in_group / gsz) * B_TN;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 64, wn = (wave & 3) * 64;
  const int srow = tid >> 2, sslot = tid & 3;
  const int sw = (sslot ^ ((srow >> 2) & 3)) * 16;
  const size_t aoffg = (size_t)min(m0 + srow, g.M - 1) * g.lda + sslot * 8;
  size_t woffg[2];
#pragma unroll
  for (int i = 0; i < 2; i++) woffg[i] = (size_t)min(n0 + srow + 128 * i, g.N - 1) * g.ldw + sslot * 8;
  u32x4 ra[3], rw[3][2];
  auto issue = [&](int k0) {
    ra[0] = *reinterpret_cast<const u32x4 *>(g.A1 + aoffg + k0);
    ra[1] = *reinterpret_cast<const u32x4 *>(g.A2 + aoffg + k0);
    ra[2] = *reinterpret_cast<const u32x4 *>(g.A3 + aoffg + k0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      rw[0][i] = *reinterpret_cast<const u32x4 *>(g.W1 + woffg[i] + k0);
      rw[1][i] = *reinterpret_cast<const u32x4 *>(g.W2 + woffg[i] + k0);
      rw[2][i] = *reinterpret_cast<const u32x4 *>(g.W3 + woffg[i] + k0);
    }
  };
  auto store = [&](int buf) {
    unsigned char *base = smem + buf * B_STAGE;
#pragma unroll
    for (int p = 0; p < 3; p++) {
      *reinterpret_cast<u32x4 *>(base + p * B_APLANE + srow * 64 + sw) = ra[p];
#pragma unroll
      for (int i = 0; i < 2; i++) *reinterpret_cast<u32x4 *>(base + 3 * B_APLANE + p * B_WPLANE + (srow + 128 * i) * 64 + sw) = rw[p][i];   // (srow + 128) >> 2 & 3 == srow >> 2 & 3
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
  const int lsw = (lhalf ^ ((lrow >> 2) & 3)) * 16;     // slot of sub-step 0; sub-step 1 is this ^ 32
  const int aoff = (wm + lrow) * 64 + lsw, boff = 3 * B_APLANE + (wn + lrow) * 64 + lsw;
  struct Frag { bf16x8 a[3][2], b[3][2]; };
  auto read_frag = [&](Frag &f, int buf, int s) {
    const unsigned char *base = smem + buf * B_STAGE;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int i = 0; i < 2; i++) {
        f.a[p][i] = *reinterpret_cast<const bf16x8 *>(base + p * B_APLANE + ((aoff + i * 32 * 64) ^ (s * 32)));
        f.b[p][i] = *reinterpret_cast<const bf16x8 *>(base + p * B_WPLANE + ((boff + i * 32 * 64) ^ (s * 32)));
      }
  };
  auto mfma_frag = [&](const Frag &f) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2][i], f.b[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][i], f.b[2][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][i], f.b[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][i], f.b[0][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][i], f.b[1][j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][i], f.b[0][j], acc[i][j], 0, 0, 0);
      }
  };
  const int nk = g.K / 32;
  Frag f0, f1;
  issue(0);
  store(0);
  if (nk > 1) issue(32);
  LDS_BARRIER();
  read_frag(f0, 0, 0);
  long long t_work = 0, t_bar = 0, t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
  // one K slice: [LDS writes of slice kt+1 | global loads of slice kt+2 | fragment reads of sub-step 1] interleaved one per MFMA
  // with the 24 MFMAs of sub-step 0; barrier; the next slice's first fragments interleaved with the 24 MFMAs of sub-step 1
  auto stage = [&](int kt, auto store_c, auto issue_c) {
    constexpr bool ST = decltype(store_c)::value, IS = decltype(issue_c)::value;
    const int cur = kt & 1, nxt = cur ^ 1;
    if constexpr (ST) store(nxt);
    if constexpr (IS) issue((kt + 2) * 32);
    read_frag(f1, cur, 1);
    mfma_frag(f0);
#ifndef NO_SCHED
    if constexpr (ST) {
#pragma unroll
      for (int i = 0; i < 9; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
      }
#pragma unroll
      for (int i = 0; i < 3; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (IS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      }
      if constexpr (IS) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 12; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (ST) read_frag(f0, nxt, 0);
    mfma_frag(f1);
#ifndef NO_SCHED
    if constexpr (ST) {
#pragma unroll
      for (int i = 0; i < 12; i++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
      }
    }
#endif
  };
  using T = std::true_type;
  using F = std::false_type;
  int kt = 0;
  for (; kt + 2 < nk; ++kt) stage(kt, T{}, T{});
  if (kt + 1 < nk) { stage(kt, T{}, F{}); ++kt; }
  stage(kt, F{}, F{});
  if (g.stamps && threadIdx.x == 0) {
    long long *q = g.stamps + (size_t)blockIdx.x * 8;
    q[0] = t_work;
    q[1] = t_bar;
    q[2] = __builtin_amdgcn_s_memtime() - t_begin;
    q[3] = __builtin_amdgcn_s_memrealtime() - r_begin;
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int n = n0 + wn + j * 32 + lrow;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhalf;
        if (m < g.M && n < g.N) g.C[(size_t)m * g.ldc + n] = acc[i][j][e];
      }
    }
}

static float frand(unsigned &s) {   // roughly normal
  float a = 0.f;
  for (int i = 0; i < 4; i++) {
    s = s * 1664525u + 1013904223u;
    a += (float)(s >> 8) / 16777216.f - 0.5f;
  }
  return a * 1.7320508f;
}

int main() {
  hipFuncSetAttribute((const void *)gemm_split3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * S_STAGE);
  hipFuncSetAttribute((const void *)gemm_split3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * S_STAGE);
  hipFuncSetAttribute((const void *)gemm_split3b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * B_STAGE);
  // ---- accuracy ----
  {
    const int M = 256, N = 256, K = 4096;
    std::vector<float> A((size_t)M * K), W((size_t)N * K), C((size_t)M * N);
    unsigned s = 12345;
    for (auto &v : A) v = frand(s);
    for (auto &v : W) v = frand(s) * 0.03f;
    float *dA, *dW, *dC;
    unsigned short *p1, *p2, *p3;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, C.size() * 4);
    hipMalloc(&p1, W.size() * 2); hipMalloc(&p2, W.size() * 2); hipMalloc(&p3, W.size() * 2);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    split_planes_kernel<<<256, 256>>>(dW, p1, p2, p3, W.size() / 2);
    SArgs g{dA, p1, p2, p3, dC, M, N, K, K, K, N, 0, nullptr};
    gemm_split3_kernel<true><<<((M + 127) / 128) * ((N + 127) / 128), 512, 2 * S_STAGE>>>(g);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double e_split = 0, e_chain = 0, m_split = 0, m_chain = 0, ref2 = 0;
    for (int m = 0; m < M; m++)
      for (int n = 0; n < N; n++) {
        double r = 0;
        float c = 0.f;
        for (int k = 0; k < K; k++) {
          r += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
          c = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], c);
        }
        const double ds = C[(size_t)m * N + n] - r, dc = c - r;
        e_split += ds * ds; e_chain += dc * dc; ref2 += r * r;
        m_split = std::max(m_split, fabs(ds)); m_chain = std::max(m_chain, fabs(dc));
      }
    const double rms = sqrt(ref2 / M / N);
    printf("accuracy K=%d (terms %d, sep_acc %d): rms(ref) %.4g | split3 rms err %.3g max %.3g | fp32 chain rms err %.3g max %.3g  (in units of rms(ref): %.3g vs %.3g)\n",
           K, SPLIT_TERMS, SEP_ACC, rms, sqrt(e_split / M / N), m_split, sqrt(e_chain / M / N), m_chain, sqrt(e_split / M / N) / rms, sqrt(e_chain / M / N) / rms);
    {   // variant B on the same data
      unsigned short *a1, *a2, *a3;
      hipMalloc(&a1, A.size() * 2); hipMalloc(&a2, A.size() * 2); hipMalloc(&a3, A.size() * 2);
      split_planes_kernel<<<256, 256>>>(dA, a1, a2, a3, A.size() / 2);
      BArgs b{a1, a2, a3, p1, p2, p3, dC, M, N, K, K, K, N, nullptr};
      hipMemset(dC, 0, C.size() * 4);
      gemm_split3b_kernel<<<((M + B_TM - 1) / B_TM) * ((N + B_TN - 1) / B_TN), 512, 2 * B_STAGE>>>(b);
      hipError_t e2 = hipDeviceSynchronize();
      if (e2 != hipSuccess) { printf("variant B launch failed: %s\n", hipGetErrorString(e2)); return 1; }
      std::vector<float> C2(C.size());
      hipMemcpy(C2.data(), dC, C2.size() * 4, hipMemcpyDeviceToHost);
      double md = 0;
      for (size_t i = 0; i < C.size(); i++) md = std::max(md, (double)fabsf(C2[i] - C[i]));
      printf("variant B vs variant A: max |diff| %.3g\n", md);
      hipFree(a1); hipFree(a2); hipFree(a3);
    }
    hipFree(dA); hipFree(dW); hipFree(dC); hipFree(p1); hipFree(p2); hipFree(p3);
  }
  // ---- speed ----
  struct Shape { int M, N, K; } shapes[] = {{9248, 4096, 1024}, {9248, 1024, 4096}, {9248, 3072, 1024}, {17440, 3072, 768}, {8192, 768, 3072}, {9248, 1024, 1024}, {4096, 4096, 4096}};
  for (auto sh : shapes) {
    float *dA, *dW, *dC;
    unsigned short *p1, *p2, *p3;
    const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K;
    hipMalloc(&dA, na * 4); hipMalloc(&dW, nw * 4); hipMalloc(&dC, (size_t)sh.M * sh.N * 4);
    hipMalloc(&p1, nw * 2); hipMalloc(&p2, nw * 2); hipMalloc(&p3, nw * 2);
    std::vector<float> h(std::max(na, nw));
    unsigned s = 777;
    for (auto &v : h) v = frand(s);
    hipMemcpy(dA, h.data(), na * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, h.data(), nw * 4, hipMemcpyHostToDevice);
    split_planes_kernel<<<1024, 256>>>(dW, p1, p2, p3, nw / 2);
    long long *dst;
    hipMalloc(&dst, 8 * 8 * 4096 * 2);
    SArgs g{dA, p1, p2, p3, dC, sh.M, sh.N, sh.K, sh.K, sh.K, sh.N, 0, dst};
    const int grid = ((sh.M + 127) / 128) * ((sh.N + 127) / 128);
    for (int spec = 1; spec < 2; spec++) {
      g.diag = spec - 1;
      auto run = [&]() {
        if (spec) gemm_split3_kernel<true><<<grid, 512, 2 * S_STAGE>>>(g);
        else gemm_split3_kernel<false><<<grid, 256, 2 * S_STAGE>>>(g);
      };
      for (int i = 0; i < 5; i++) run();
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int i = 0; i < 20; i++) run();
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20;
      {
        std::vector<long long> st((size_t)grid * 8);
        hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
        double w[2] = {0, 0}, b[2] = {0, 0}, life = 0, real = 0;
        for (int i = 0; i < grid; i++) {
          for (int r = 0; r < 2; r++) { w[r] += st[i * 8 + r * 4]; b[r] += st[i * 8 + r * 4 + 1]; }
          life += st[i * 8 + 2]; real += st[i * 8 + 3];
        }
        printf("   per K slice (cycles): consumer work %.0f barrier %.0f | producer work %.0f barrier %.0f | clock %.2f GHz\n", w[0] / grid / (sh.K / 32), b[0] / grid / (sh.K / 32),
               w[1] / grid / (sh.K / 32), b[1] / grid / (sh.K / 32), life / real * 0.1);
      }
      printf("(%d,%d,%d) diag+1=%d: %.1f us  = %.1f TFLOP/s fp32-equivalent (bf16 MFMA rate %.0f TFLOP/s)\n", sh.M, sh.N, sh.K, spec, ms * 1e3,
             2.0 * sh.M * sh.N * sh.K / ms / 1e9, 2.0 * SPLIT_TERMS * sh.M * sh.N * sh.K / ms / 1e9);
    }
    {
      unsigned short *a1, *a2, *a3;
      hipMalloc(&a1, na * 2); hipMalloc(&a2, na * 2); hipMalloc(&a3, na * 2);
      split_planes_kernel<<<1024, 256>>>(dA, a1, a2, a3, na / 2);
      BArgs b{a1, a2, a3, p1, p2, p3, dC, sh.M, sh.N, sh.K, sh.K, sh.K, sh.N, dst};
      const int gb = ((sh.M + B_TM - 1) / B_TM) * ((sh.N + B_TN - 1) / B_TN);
      for (int i = 0; i < 5; i++) gemm_split3b_kernel<<<gb, 512, 2 * B_STAGE>>>(b);
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      for (int i = 0; i < 20; i++) gemm_split3b_kernel<<<gb, 512, 2 * B_STAGE>>>(b);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= 20;
      std::vector<long long> st((size_t)gb * 8);
      hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
      double w = 0, bq = 0, life = 0, real = 0;
      for (int i = 0; i < gb; i++) { w += st[i * 8]; bq += st[i * 8 + 1]; life += st[i * 8 + 2]; real += st[i * 8 + 3]; }
      // split pass time
      hipEventRecord(e0);
      for (int i = 0; i < 10; i++) split_planes_kernel<<<2048, 256>>>(dA, a1, a2, a3, na / 2);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms2;
      hipEventElapsedTime(&ms2, e0, e1);
      printf("(%d,%d,%d) variant B: %.1f us = %.1f TFLOP/s fp32-equivalent (bf16 rate %.0f) | per K slice: work %.0f barrier %.0f cycles (floor 1536), clock %.2f GHz | A split pass %.1f us\n",
             sh.M, sh.N, sh.K, ms * 1e3, 2.0 * sh.M * sh.N * sh.K / ms / 1e9, 12.0 * sh.M * sh.N * sh.K / ms / 1e9, w / gb / (sh.K / 32), bq / gb / (sh.K / 32), life / real * 0.1, ms2 * 100);
      hipFree(a1); hipFree(a2); hipFree(a3);
    }
    hipFree(dA); hipFree(dW); hipFree(dC); hipFree(p1); hipFree(p2); hipFree(p3);
  }
  return 0;
}
