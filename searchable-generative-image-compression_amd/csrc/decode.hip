// Decode-side HBM-bound kernels (generative decoder path): GroupNorm(+swish) with
// optional zero-halo output for the implicit-GEMM 3x3 convs, nearest-2x upsample into a halo buffer, halo
// copy, row softmax, PixelShuffle into the tile-major feature layout, decoder token assembly, codebook
// gather + l2-norm, and the final clamp + NHWC->NCHW.
// Reference: taming/modules/diffusionmodules/model.py:34-53,78-192,506-537; models/codec_sq_fixbpp.py:203-207,
// 248-300,658-669,886-892.
#include "common.h"
#include "split3.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
static inline unsigned ew_grid(long n) {
  long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

__device__ __forceinline__ long tm16_row(int b, int y, int x, int H, int W) {
  const int nH = H >> 4, nW = W >> 4;
  return ((((long)b * nH + (y >> 4)) * nW + (x >> 4)) << 8) + ((y & 15) << 4) + (x & 15);
}

// ------------------------------------------------------------------------------------------------
// GroupNorm(32 groups, eps) over NHWC x [B, HW, C]:
//   k1: per (b, split) partial per-channel (sum, sumsq) in fp64      [B, S, C, 2]
//   k2: per (b, group) mean / rstd                                     [B, 32, 2]
//   k3: y = (x - mean) * rstd * gamma + beta, optional swish; output plain [B,HW,C] or into the interior of
//       a zero-halo buffer [B, H+2, W+2, C] (the next 3x3 conv reads it as an implicit GEMM)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_partial_kernel(const float *__restrict__ x, int HW, int C, int S,
                                                         double *__restrict__ part) {
  // block = (b, split); thread t owns channel quad (t % C4) and pixel lane (t / C4)
  extern __shared__ double sh[];  // [256][8]
  const int b = blockIdx.x / S, sp = blockIdx.x % S;
  const int C4 = C >> 2, tid = threadIdx.x;
  const int lanes = 256 / C4;  // pixel lanes per block (C4 <= 256 and divides 256 for C in {32..1024})
  const int c4 = tid % C4, pl = tid / C4;
  const int per = (HW + S - 1) / S, p0 = sp * per, p1 = min(HW, p0 + per);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (pl < lanes) {
    // 4 independent loads in flight per thread (the accumulate chain is fp64; without unrolling the loop is
    // load-latency-bound)
    int p = p0 + pl;
    for (; p + 3 * lanes < p1; p += 4 * lanes) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = reinterpret_cast<const f32x4 *>(x + ((long)b * HW + p + u * lanes) * C)[c4];
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
          s[t] += (double)v[u][t];
          q[t] += (double)v[u][t] * (double)v[u][t];
        }
    }
    for (; p < p1; p += lanes) {
      const f32x4 v = reinterpret_cast<const f32x4 *>(x + ((long)b * HW + p) * C)[c4];
#pragma unroll
      for (int t = 0; t < 4; t++) {
        s[t] += (double)v[t];
        q[t] += (double)v[t] * (double)v[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; t++) sh[tid * 8 + t] = s[t], sh[tid * 8 + 4 + t] = q[t];
  __syncthreads();
  if (tid < C4) {  // fixed-order reduction over the pixel lanes -> deterministic
    double rs[4] = {0, 0, 0, 0}, rq[4] = {0, 0, 0, 0};
    for (int l = 0; l < lanes; l++)
#pragma unroll
      for (int t = 0; t < 4; t++) rs[t] += sh[(l * C4 + tid) * 8 + t], rq[t] += sh[(l * C4 + tid) * 8 + 4 + t];
    double *o = part + (((long)b * S + sp) * C + tid * 4) * 2;
#pragma unroll
    for (int t = 0; t < 4; t++) o[t * 2] = rs[t], o[t * 2 + 1] = rq[t];
  }
}

__global__ void gn_finalize_kernel(const double *__restrict__ part, int B, int HW, int C, int S, int G, float eps,
                                   float *__restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, g = i % G, cpg = C / G;
  double s = 0, q = 0;
  for (int sp = 0; sp < S; sp++)
    for (int c = g * cpg; c < (g + 1) * cpg; c++) {
      const double *p = part + (((long)b * S + sp) * C + c) * 2;
      s += p[0];
      q += p[1];
    }
  const double n = (double)HW * cpg, mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  stats[i * 2] = (float)mean;
  stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// grid = (B*H rows, chunks of the row); each thread owns one channel quad (fixed gamma/beta/stats) and walks x --
// no integer divisions in the loop (the flat-index version was ALU-bound on its div/mod chain)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float *__restrict__ x, const float *__restrict__ stats,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int B, int H, int W, int C, int G, int swish, int halo,
                                                       float *__restrict__ y, unsigned short *__restrict__ planes) {
  const int C4 = C >> 2, cpg = C / G;
  const int row = blockIdx.x;            // b*H + yy
  const int b = row / H, yy = row - b * H;
  const int c4 = threadIdx.x % C4, xl = threadIdx.x / C4, xstep = 256 / C4;   // C4 divides 256
  const f32x4 g4 = reinterpret_cast<const f32x4 *>(gamma)[c4], b4 = reinterpret_cast<const f32x4 *>(beta)[c4];
  float mean[4], rstd[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int g = (c4 * 4 + e) / cpg;
    mean[e] = stats[(b * G + g) * 2];
    rstd[e] = stats[(b * G + g) * 2 + 1];
  }
  const float *xr = x + (long)row * W * C;
  float *yr = halo ? y + (((long)b * (H + 2) + yy + 1) * (W + 2) + 1) * C : y + (long)row * W * C;
  for (int xx = blockIdx.y * xstep + xl; xx < W; xx += gridDim.y * xstep) {
    const f32x4 v = reinterpret_cast<const f32x4 *>(xr + (long)xx * C)[c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float r = (v[e] - mean[e]) * rstd[e] * g4[e] + b4[e];
      if (swish) r = r / (1.0f + __expf(-r));  // x * sigmoid(x)
      o[e] = r;
    }
    if (planes) {   // halo output as slice-major bf16x3 planes [3][C / 32][B (H+2) (W+2)][32]: the operand of an implicit split-GEMM convolution
      const long prow = ((long)b * (H + 2) + yy + 1) * (W + 2) + 1 + xx;
      s3_store4(planes, (long)B * (H + 2) * (W + 2) * C, s3_pack_off(prow, c4 * 4, (size_t)B * (H + 2) * (W + 2)), o);
    } else {
      reinterpret_cast<f32x4 *>(yr + (long)xx * C)[c4] = o;
    }
  }
}

static int groupnorm_any(const float *d_x, const float *d_gamma, const float *d_beta, int B, int H, int W, int C, int groups,
                         float eps, int swish, int halo_out, double *d_ws, float *d_stats, float *d_y, unsigned short *d_planes,
                         sgic_stream_t stream);

extern "C" int sgic_groupnorm_nhwc(const float *d_x, const float *d_gamma, const float *d_beta, int B, int H, int W, int C,
                                   int groups, float eps, int swish, int halo_out, double *d_ws, float *d_stats,
                                   float *d_y, sgic_stream_t stream) {
  SGIC_REQUIRE(d_y, "y");
  return groupnorm_any(d_x, d_gamma, d_beta, B, H, W, C, groups, eps, swish, halo_out, d_ws, d_stats, d_y, nullptr, stream);
}

// GroupNorm (+swish) whose output feeds a 3x3 convolution run as an implicit split GEMM (sgic_conv3x3_split3_f32): the
// interior of the zero-halo planes buffer [3][B (H+2) (W+2)][C] is written directly (its border must already be zero)
extern "C" int sgic_groupnorm_nhwc_split3(const float *d_x, const float *d_gamma, const float *d_beta, int B, int H, int W, int C,
                                          int groups, float eps, int swish, double *d_ws, float *d_stats,
                                          uint16_t *d_halo_planes, sgic_stream_t stream) {
  SGIC_REQUIRE(d_halo_planes && ((uintptr_t)d_halo_planes & 7) == 0, "planes");
  return groupnorm_any(d_x, d_gamma, d_beta, B, H, W, C, groups, eps, swish, 1, d_ws, d_stats, nullptr, d_halo_planes, stream);
}

static int groupnorm_any(const float *d_x, const float *d_gamma, const float *d_beta, int B, int H, int W, int C, int groups,
                         float eps, int swish, int halo_out, double *d_ws, float *d_stats, float *d_y, unsigned short *d_planes,
                         sgic_stream_t stream) {
  SGIC_REQUIRE(d_x && d_gamma && d_beta && d_ws && d_stats && (d_y || d_planes) && B > 0 && H > 0 && W > 0, "args");
  SGIC_REQUIRE(C % 4 == 0 && C % groups == 0 && (C / 4) <= 256 && 256 % (C / 4) == 0, "C must be 4*2^k <= 1024 and a multiple of groups");
  const int HW = H * W;
  int S = HW / 1024;
  S = S < 1 ? 1 : (S > 64 ? 64 : S);  // d_ws must hold B*64*C*2 doubles
  hipStream_t st = to_stream(stream);
  gn_partial_kernel<<<B * S, 256, 256 * 8 * sizeof(double), st>>>(d_x, HW, C, S, d_ws);
  gn_finalize_kernel<<<cdiv((size_t)B * groups, 64), 64, 0, st>>>(d_ws, B, HW, C, S, groups, eps, d_stats);
  {
    const int xstep = 256 / (C / 4);
    int chunks = (W + xstep * 8 - 1) / (xstep * 8);   // ~8 positions per thread
    chunks = chunks < 1 ? 1 : chunks;
    gn_apply_kernel<<<dim3(B * H, chunks), 256, 0, st>>>(d_x, d_stats, d_gamma, d_beta, B, H, W, C, groups, swish, halo_out, d_y, d_planes);
  }
  return sgic::check_launch("groupnorm");
}

// ------------------------------------------------------------------------------------------------
// Halo copies: out [B, OH+2, OW+2, C] interior = in (plain NHWC or TM16 rows), optionally nearest-2x
// upsampled (Upsample, model.py:49-53).  The halo itself is zeroed once by the caller (memset).
// ------------------------------------------------------------------------------------------------
__global__ void halo_copy_kernel(const float *__restrict__ in, int B, int H, int W, int C, int up, int tile16,
                                 float *__restrict__ out, unsigned short *__restrict__ planes) {
  const int C4 = C >> 2, OH = H << up, OW = W << up;
  const long total = (long)B * OH * OW * C4;
  GRID_STRIDE(i, total) {
    const int c4 = (int)(i % C4);
    long t = i / C4;
    const int ox = (int)(t % OW);
    t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    const int sy = oy >> up, sx = ox >> up;
    const long irow = tile16 ? tm16_row(b, sy, sx, H, W) : ((long)b * H + sy) * W + sx;
    const f32x4 v = reinterpret_cast<const f32x4 *>(in + irow * C)[c4];
    const long prow = ((long)b * (OH + 2) + oy + 1) * (OW + 2) + ox + 1;
    if (planes) s3_store4(planes, (long)B * (OH + 2) * (OW + 2) * C, s3_pack_off(prow, c4 * 4, (size_t)B * (OH + 2) * (OW + 2)), v);
    else reinterpret_cast<f32x4 *>(out + prow * C)[c4] = v;
  }
}

extern "C" int sgic_halo_copy(const float *d_in, int B, int H, int W, int C, int upsample2x, int tile16, float *d_out,
                              sgic_stream_t stream) {
  SGIC_REQUIRE(d_in && d_out && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "args");
  SGIC_REQUIRE(!tile16 || (H % 16 == 0 && W % 16 == 0), "tile-major input needs H,W multiples of 16");
  const int up = upsample2x ? 1 : 0;
  halo_copy_kernel<<<ew_grid((long)B * (H << up) * (W << up) * C / 4), 256, 0, to_stream(stream)>>>(d_in, B, H, W, C, up,
                                                                                                   tile16, d_out, nullptr);
  return sgic::check_launch("halo_copy_kernel");
}

// the same copy into the interior of a zero-halo bf16x3 planes buffer [3][B (OH+2) (OW+2)][C] (operand of
// sgic_conv3x3_split3_f32)
extern "C" int sgic_halo_copy_split3(const float *d_in, int B, int H, int W, int C, int upsample2x, int tile16,
                                     uint16_t *d_halo_planes, sgic_stream_t stream) {
  SGIC_REQUIRE(d_in && d_halo_planes && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && ((uintptr_t)d_halo_planes & 7) == 0, "args");
  SGIC_REQUIRE(!tile16 || (H % 16 == 0 && W % 16 == 0), "tile-major input needs H,W multiples of 16");
  const int up = upsample2x ? 1 : 0;
  halo_copy_kernel<<<ew_grid((long)B * (H << up) * (W << up) * C / 4), 256, 0, to_stream(stream)>>>(d_in, B, H, W, C, up,
                                                                                                   tile16, nullptr, d_halo_planes);
  return sgic::check_launch("halo_copy_kernel");
}

// ------------------------------------------------------------------------------------------------
// Row softmax  y = softmax(scale * x)  over rows of length L (one wave per row; L <= 4096)
// (AttnBlock model.py:181-183; soft codebook lookup codec_sq_fixbpp.py:660-661)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float *__restrict__ x, float *__restrict__ y, long M, int L,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const float *xp = x + m * L;
  float *yp = y + m * L;
  float mx = -INFINITY;
  for (int c = lane; c < L; c += 64) mx = fmaxf(mx, xp[c] * scale);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float s = 0.f;
  for (int c = lane; c < L; c += 64) s += expf(xp[c] * scale - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / s;
  for (int c = lane; c < L; c += 64) yp[c] = expf(xp[c] * scale - mx) * inv;
}

extern "C" int sgic_softmax_rows(const float *d_x, float *d_y, long M, int L, float scale, sgic_stream_t stream) {
  SGIC_REQUIRE(d_x && d_y && M > 0 && L > 0, "args");
  softmax_rows_kernel<<<cdiv(M, 4), 256, 0, to_stream(stream)>>>(d_x, d_y, M, L, scale);
  return sgic::check_launch("softmax_rows_kernel");
}

// ------------------------------------------------------------------------------------------------
// PixelShuffle(2) of a [(b,y,x) plain, 4C] map into the TM16 feature layout [(b, 2y+i, 2x+j), C]:
// out[.., c] = in[.., c*4 + i*2 + j]   (nn.PixelShuffle semantics; codec_sq_fixbpp.py:203-207)
// ------------------------------------------------------------------------------------------------
__global__ void pixel_shuffle_kernel(const float *__restrict__ in, int B, int H, int W, int C, float *__restrict__ out) {
  const long total = (long)B * H * W * 4 * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ij = (int)(t & 3);
    t >>= 2;
    const int x = (int)(t % W);
    t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const float v = in[(((long)b * H + y) * W + x) * 4 * C + c * 4 + ij];
    out[tm16_row(b, 2 * y + (ij >> 1), 2 * x + (ij & 1), 2 * H, 2 * W) * C + c] = v;
  }
}

extern "C" int sgic_pixel_shuffle2_tm16(const float *d_in, int B, int H, int W, int C, float *d_out, sgic_stream_t stream) {
  SGIC_REQUIRE(d_in && d_out && B > 0 && H > 0 && W > 0 && C > 0 && (2 * H) % 16 == 0 && (2 * W) % 16 == 0, "args");
  pixel_shuffle_kernel<<<ew_grid((long)B * H * W * 4 * C), 256, 0, to_stream(stream)>>>(d_in, B, H, W, C, d_out);
  return sgic::check_launch("pixel_shuffle_kernel");
}

// ------------------------------------------------------------------------------------------------
// Decoder token assembly (codec_sq_fixbpp.py:258-267):
//   out[n,0] = cls + pos[0];  out[n,1+p] = mask + pos[1+p];  out[n,1+P+t] = emb[n*T+t] + latpos[t]
// ------------------------------------------------------------------------------------------------
__global__ void assemble_dec_tokens_kernel(const float *__restrict__ emb, const float *__restrict__ cls,
                                           const float *__restrict__ mask, const float *__restrict__ pos,
                                           const float *__restrict__ latpos, int N, int P, int T, int D,
                                           float *__restrict__ out) {
  const int L = 1 + P + T, D4 = D >> 2;
  const long total = (long)N * L * D4;
  GRID_STRIDE(i, total) {
    const int d4 = (int)(i % D4);
    const long r = i / D4;
    const int l = (int)(r % L);
    const int n = (int)(r / L);
    f32x4 a, b;
    if (l == 0) {
      a = reinterpret_cast<const f32x4 *>(cls)[d4];
      b = reinterpret_cast<const f32x4 *>(pos)[d4];
    } else if (l <= P) {
      a = reinterpret_cast<const f32x4 *>(mask)[d4];
      b = reinterpret_cast<const f32x4 *>(pos + (long)l * D)[d4];
    } else {
      a = reinterpret_cast<const f32x4 *>(emb + ((long)n * T + (l - 1 - P)) * D)[d4];
      b = reinterpret_cast<const f32x4 *>(latpos + (long)(l - 1 - P) * D)[d4];
    }
    reinterpret_cast<f32x4 *>(out + r * D)[d4] = a + b;
  }
}

extern "C" int sgic_assemble_dec_tokens(const float *d_emb, const float *d_cls, const float *d_mask, const float *d_pos,
                                        const float *d_latpos, int N, int P, int T, int D, float *d_out,
                                        sgic_stream_t stream) {
  SGIC_REQUIRE(d_emb && d_cls && d_mask && d_pos && d_latpos && d_out && N > 0 && P > 0 && T > 0 && (D & 3) == 0, "args");
  assemble_dec_tokens_kernel<<<ew_grid((long)N * (1 + P + T) * D / 4), 256, 0, to_stream(stream)>>>(d_emb, d_cls, d_mask, d_pos,
                                                                                                    d_latpos, N, P, T, D, d_out);
  return sgic::check_launch("assemble_dec_tokens_kernel");
}

// ------------------------------------------------------------------------------------------------
// z_hat: codebook rows gathered by index, l2-normalised over the code dimension
// (codec_sq_fixbpp.py:889-892).  out [(n,t), dim] padded to ld floats (zeros) so it can feed the GEMM.
// ------------------------------------------------------------------------------------------------
__global__ void codebook_gather_norm_kernel(const int *__restrict__ idx, const float *__restrict__ cb, int M, int dim, int ld,
                                            float *__restrict__ out) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float *e = cb + (long)idx[m] * dim;
  float s = 0.f;
  for (int d = 0; d < dim; d++) s += e[d] * e[d];
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  for (int d = 0; d < ld; d++) out[(long)m * ld + d] = d < dim ? e[d] * inv : 0.f;
}

extern "C" int sgic_codebook_gather_norm(const int32_t *d_idx, const float *d_codebook, int M, int dim, int ld, float *d_out,
                                         sgic_stream_t stream) {
  SGIC_REQUIRE(d_idx && d_codebook && d_out && M > 0 && dim > 0 && ld >= dim, "args");
  codebook_gather_norm_kernel<<<cdiv(M, 64), 64, 0, to_stream(stream)>>>(d_idx, d_codebook, M, dim, ld, d_out);
  return sgic::check_launch("codebook_gather_norm_kernel");
}

// ------------------------------------------------------------------------------------------------
// x_hat: [(b,y,x), ldc >= 3] -> clamp(-1,1) -> NCHW (B,3,H,W)   (codec_sq_fixbpp.py:901)
// ------------------------------------------------------------------------------------------------
__global__ void nhwc3_to_nchw_clamp_kernel(const float *__restrict__ in, int ld, int B, int H, int W, float *__restrict__ out) {
  const long total = (long)B * 3 * H * W;
  GRID_STRIDE(i, total) {
    const int x = (int)(i % W);
    long t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % 3);
    const int b = (int)(t / 3);
    const float v = in[(((long)b * H + y) * W + x) * ld + c];
    out[i] = fminf(fmaxf(v, -1.f), 1.f);
  }
}

extern "C" int sgic_nhwc3_to_nchw_clamp(const float *d_in, int ld, int B, int H, int W, float *d_out, sgic_stream_t stream) {
  SGIC_REQUIRE(d_in && d_out && ld >= 3 && B > 0 && H > 0 && W > 0, "args");
  nhwc3_to_nchw_clamp_kernel<<<ew_grid((long)B * 3 * H * W), 256, 0, to_stream(stream)>>>(d_in, ld, B, H, W, d_out);
  return sgic::check_launch("nhwc3_to_nchw_clamp_kernel");
}

// ------------------------------------------------------------------------------------------------
// Exact top-k per row (descending score, ties -> lower index) for IndexFlatIP.search semantics
// (search.py:113-120): scores [nq, n] come from sgic_gemm_f32(q, db).  One workgroup per query row,
// k selection passes; k <= 1024, n arbitrary.  Scores are overwritten with -inf as they are taken.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_rows_kernel(float *__restrict__ scores, int n, int k, float *__restrict__ out_s,
                                                        int *__restrict__ out_i) {
  __shared__ float sv[256];
  __shared__ int si[256];
  float *row = scores + (long)blockIdx.x * n;
  for (int j = 0; j < k; j++) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < n; c += 256) {
      const float v = row[c];
      if (v > best || (v == best && c < bi)) best = v, bi = c;
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const float ov = sv[threadIdx.x + s];
        const int oi = si[threadIdx.x + s];
        if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) sv[threadIdx.x] = ov, si[threadIdx.x] = oi;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      out_s[(long)blockIdx.x * k + j] = sv[0];
      out_i[(long)blockIdx.x * k + j] = si[0] == 0x7fffffff ? -1 : si[0];
      if (si[0] != 0x7fffffff) row[si[0]] = -INFINITY;
    }
    __syncthreads();
  }
}

extern "C" int sgic_topk_rows(float *d_scores, int nq, int n, int k, float *d_out_scores, int32_t *d_out_idx,
                              sgic_stream_t stream) {
  SGIC_REQUIRE(d_scores && d_out_scores && d_out_idx && nq > 0 && n > 0 && k > 0 && k <= 1024 && k <= n, "args");
  topk_rows_kernel<<<nq, 256, 0, to_stream(stream)>>>(d_scores, n, k, d_out_scores, d_out_idx);
  return sgic::check_launch("topk_rows_kernel");
}
