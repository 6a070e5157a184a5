/*
 * sgic.h -- C ABI of libsgic.so, the MI355X (gfx950) hot path of the searchable generative image codec.
 *
 * Every entry point is `extern "C"`, takes plain pointers + sizes (no torch / pybind types), returns
 * 0 on success or a negative SGIC_E* code, never throws, and launches on the HIP stream it is given
 * (pass NULL for the default stream).  Pointers named d_* are DEVICE pointers (HBM); everything else is
 * host memory.  Handles are thread-compatible: one handle per thread/stream; the library has no mutable
 * process-global launch state (launch options travel per call in sgic_launch_opts).
 *
 * Each block below cites the reference interface (file:line under /root/reference/src) it replaces;
 * INTEGRATION.md shows the reference-side binding.
 */
#ifndef SGIC_H
#define SGIC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t *sgic_stream_t; /* == hipStream_t */

#define SGIC_OK 0
#define SGIC_EINVAL -1  /* bad argument / shape the kernel does not support */
#define SGIC_EHIP -2    /* a HIP runtime call failed (see sgic_last_error) */
#define SGIC_ENOSPC -3  /* output buffer too small */
#define SGIC_ENODEV -4  /* no gfx950 device visible */

const char *sgic_last_error(void);
int sgic_version(void);
/* number of visible HIP devices, or SGIC_ENODEV; does not create a context */
int sgic_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Entropy coder  (replaces entropy.MLCodec_rans / MLCodec_CXX: cpp/py_rans/py_rans.h:13-57,
 * cpp/py_rans/py_rans.cpp:22-221, cpp/rans/rans.cpp:71-187,280-362, cpp/ops/ops.cpp:24-82)
 * ------------------------------------------------------------------------------------------- */

/* MLCodec_CXX.pmf_to_quantized_cdf(pmf, precision) -> cdf[n+1]   (ops.cpp:24-82).  Host-side table
 * builder, runs once per process at update() time. */
int sgic_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf_out);

/* CDF group = what RansEncoder/RansDecoder.add_cdf registers (py_rans.cpp:47-76, rans.cpp:71-93,
 * 287-295).  cdf is (rows, cols) int32 row-major on the HOST; it is uploaded to HBM once. */
typedef struct sgic_cdf_table sgic_cdf_table;
int sgic_cdf_table_create(const int32_t *cdf, int rows, int cols, const int32_t *sizes, const int32_t *offsets,
                          sgic_cdf_table **out);
void sgic_cdf_table_destroy(sgic_cdf_table *t);

/* Batched RansEncoder.{reset, encode_with_indexes x K, flush, get_encoded_stream} for B independent
 * images (py_rans.cpp:22-45,85-136; rans.cpp:101-187).  d_sym/d_idx: (B, n_per_img) int16, for each
 * image the concatenation of all encode_with_indexes calls in call order; idx < 0 => symbol skipped.
 * Stream b is written END-ALIGNED inside its slot: bytes d_out[b*cap + d_off[b] .. b*cap + cap),
 * d_len[b] = cap - d_off[b]; byte 0 of the stream is the 0x01 single-stream flag.  d_err[b] != 0 if
 * the slot was too small (SGIC_ENOSPC) or an index was out of range (SGIC_EINVAL). */
int sgic_rans_encode_batch(const sgic_cdf_table *t, const int16_t *d_sym, const int16_t *d_idx, int B,
                           int n_per_img, uint8_t *d_out, int cap, int32_t *d_off, int32_t *d_len,
                           int32_t *d_err, sgic_stream_t stream);

/* RansDecoder.set_stream for B images (py_rans.cpp:150-185, rans.cpp:280-285).  Streams are
 * (B, cap) with d_off/d_len as produced above (or d_off = 0 for front-aligned uploads).
 * d_state: (B, 4) uint32 cursor {x, pos, err, _}. */
int sgic_rans_decode_init_batch(const uint8_t *d_streams, int cap, const int32_t *d_off, const int32_t *d_len,
                                int B, uint32_t *d_state, sgic_stream_t stream);
/* RansDecoder.decode_stream (rans.cpp:303-362): continues from the cursor.  Image b reads its indexes at
 * d_idx + b*idx_stride (n int16) and writes symbols at d_sym_out + b*out_stride (0 where idx<0), so a
 * step slice of a (B,4,n) buffer can be addressed in place.  Reads past the end of a stream are
 * bounds-checked: the state's err word is set instead of over-reading like the reference. */
int sgic_rans_decode_batch(const sgic_cdf_table *t, const uint8_t *d_streams, int cap, const int32_t *d_off,
                           const int32_t *d_len, int B, uint32_t *d_state, const int16_t *d_idx, int n,
                           int idx_stride, int16_t *d_sym_out, int out_stride, sgic_stream_t stream);

/* z-branch: torchac.encode_float_cdf / decode_float_cdf with the uniform 4096-symbol cdf
 * (models/codec_sq_fixbpp.py:841-846,863-864,886-887) == 12-bit MSB-first packing + "01" terminator.
 * d_idx: (B, n) int32 indices in [0,4096); d_out: (B, sgic_pack12_size(n)) bytes. */
size_t sgic_pack12_size(size_t n);
int sgic_pack12_batch(const int32_t *d_idx, int B, int n, uint8_t *d_out, sgic_stream_t stream);
int sgic_unpack12_batch(const uint8_t *d_in, int B, int n, int32_t *d_idx, sgic_stream_t stream);

/* One step k (0..3) of the 4-step masked quantiser fused with build_indexes
 * (entropy/compression_model.py:224-239,296-366; entropy/entropy_models.py:355-362,66-69).
 * Layout NHWC: d_y (B*HW, 64) already divided by clamp_min(q_step,0.5); d_scales/d_means rows of
 * ld_sm floats; d_yhat (B*HW rows, ld_yhat floats): the active (channel,pos) entries get y_q + mean.
 * d_sym/d_idx: (B, 4, 16, H, W) int16, step k slice written.  thr < 0 disables force-zero/skip. */
int sgic_quant_step(const float *d_y, const float *d_scales, const float *d_means, int ld_sm, float *d_yhat,
                    int ld_yhat, int B, int H, int W, int C, int k, float thr, int16_t *d_sym, int16_t *d_idx,
                    sgic_stream_t stream);
/* GaussianEncoder.build_indexes on a flat array (entropy/entropy_models.py:355-362); thr < 0: no skip marking. */
int sgic_scale_to_index(const float *d_scales, long n, float thr, int16_t *d_idx, sgic_stream_t stream);
/* Decoder twins (compression_model.py:377-418): indexes from scales for step k, and
 * y_hat[active] = sym + mean after the symbols were decoded. */
int sgic_index_step(const float *d_scales, int ld_sm, int B, int H, int W, int C, int k, float thr,
                    int16_t *d_idx, sgic_stream_t stream);
/* Decision margins of sgic_index_step for step k (robust decoding of streams written by another fp32 implementation):
 * d_margin (B,4,C/4,H,W) float = distance of each coded sigma, in index steps, to the nearest boundary of build_indexes
 * (a bin edge or the skip threshold; entropy_models.py:355-362), d_alt = the index on the other side of that boundary. */
int sgic_index_margins(const float *d_scales, int ld_sm, int B, int H, int W, int C, int k, float thr, float *d_margin,
                       int16_t *d_alt, sgic_stream_t stream);
int sgic_dequant_step(const int16_t *d_sym, const float *d_means, int ld_sm, float *d_yhat, int ld_yhat, int B,
                      int H, int W, int C, int k, sgic_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Transform kernels (fp32, row-major "token x channel" = NHWC activations).  They replace the stock
 * ATen ops under the reference's nn.Modules; reference call sites are cited per function.
 * ------------------------------------------------------------------------------------------- */
#define SGIC_ACT_NONE 0
#define SGIC_ACT_GELU 1  /* exact erf GELU (nn.GELU default) */
#define SGIC_ACT_SILU 2
#define SGIC_ACT_TANH 3
#define SGIC_ACT_LRELU 4 /* LeakyReLU(0.01) */

/* Launch options of the GEMM-family and attention entry points.  Passed per call (NULL = defaults): the library keeps
 * NO process-global launch state, so two threads / streams can launch with different options concurrently.
 *   tile_mode (sgic_gemm_f32 / sgic_gemm_batched_f32 / sgic_conv3x3_f32): 0 = built-in heuristic, 1 = 128x128,
 *     2 = 128x64 workgroup tiles (double-buffered LDS), 3 / 4 = the same tiles with a single LDS buffer (3-4 workgroups
 *     per CU), 5..8 = 1..4 with a start-up stagger of the co-resident workgroups, 9 / 10 = MIXED: whole rounds of
 *     128x128 tiles for the bulk of the rows + 64x64 tiles for the remaining rows in the same launch (2 / 1 LDS
 *     buffers), 11 = persistent 128x128 (2 resident workgroups per CU walk the tile list), 12 = persistent MIXED,
 *     13 / 14 = 64x64 tiles as their own launch (2 / 1 buffers; small-M GEMMs), 15 = latency kernel for under-filled
 *     launches (32x32 tiles of 16x16x4-MFMA blocks: 4x shorter dependent chains, 4x the workgroups; needs K % 64 == 0,
 *     otherwise runs as 13).  Results are bitwise identical for
 *     every choice (the k order is fixed by K alone); the host autotuner (sgic_amd.ops) picks per shape and persists
 *     its picks.
 *   attn_mode (sgic_attention_f32): low 3 bits: 0 = built-in choice, 1..6 = K/V ring depth x start-up stagger, 7 = three query
 *     row blocks per workgroup with a unit's row blocks padded to a multiple of three (three workgroups per CU: two whole rounds
 *     for L = 289 at batch 32) (results bitwise identical for every choice); +8: S^T = K Q^T as a bf16x3 split product on the bf16
 *     matrix pipe (fp32-accurate, the arithmetic of sgic_gemm_split3_f32; differs from the fp32-MFMA variant in the last bits,
 *     identical across 1..7).
 *   profiler: non-NULL and open => the launch is dispatched with its own (start, stop) event pair (hipExtLaunchKernel:
 *     timestamps taken by the dispatch itself, no extra packets on the stream). */
typedef struct sgic_profiler sgic_profiler;
typedef struct sgic_launch_opts {
  int tile_mode;
  int attn_mode;
  sgic_profiler *profiler;
  int w_packed;   /* split GEMM / convolution: d_Wplanes is in the slice-major layout of sgic_split3_pack_f32 */
  int a_packed;   /* split GEMM with d_A == NULL: d_Aplanes is slice-major [3][K / 32][M][32] -- what sgic_layernorm_split3_f32,
                     sgic_attention_split3_f32, a GEMM's d_Cplanes output and sgic_split3_pack_f32 write; 0 = row-major [3][M][K]
                     (sgic_split3_f32) */
} sgic_launch_opts;

/* Profile window for the roofline figure of bench.py.  One profiler per launching thread.  begin() opens a window,
 * every launch that carries the profiler in its opts is timed, end() closes the window and returns the duration in
 * milliseconds of each launch in launch order (at most cap of them; *n_out = launches seen). */
int sgic_profiler_create(sgic_profiler **out);
void sgic_profiler_destroy(sgic_profiler *p);
int sgic_profiler_begin(sgic_profiler *p, int max_launches);
int sgic_profiler_end(sgic_profiler *p, float *ms_out, int cap, int *n_out);

/* C[M,N] = act(A[M,K] . W[N,K]^T + bias[N]) + R[M,N]   -- nn.Linear / 1x1 Conv2d / im2col'd convs
 * (titok/blocks.py:37-64, blocks/swin_transformer.py:30-39,86,92, models/cross_blocks.py:55-68,
 * blocks/conv_blocks.py:63-67, blocks/dcvc.py:17-23,42-43).  fp32-input MFMA (exact fp32), fixed k
 * order => batch-invariant.  K, lda, ldw multiples of 4; A, W 16-byte aligned; bias/R optional.
 * Row maps row(m) = (m / seg) * seg_stride + m % seg (seg = 0: identity) let A be read from / C be
 * written to a token slice [:, a:b] of an (n, L, C) buffer in place (models/cross_blocks.py:87-94). */
int sgic_gemm_f32(const float *d_A, int lda, const float *d_W, int ldw, const float *d_bias, const float *d_R,
                  int ldr, float *d_C, int ldc, int M, int N, int K, int act, int a_seg, int a_seg_stride,
                  int c_seg, int c_seg_stride, const sgic_launch_opts *opts, sgic_stream_t stream);

/* Row LayerNorm, biased variance, eps inside sqrt, optional fused SiLU (act = SGIC_ACT_SILU); rows of x
 * and y addressed through the same kind of segment map (titok/blocks.py:36,42,
 * blocks/swin_transformer.py:135,142, blocks/conv_blocks.py:62, models/cross_blocks.py:62,67). */
int sgic_layernorm_f32(const float *d_x, int ldx, int xseg, int xseg_stride, const float *d_gamma,
                       const float *d_beta, float *d_y, int ldy, int yseg, int yseg_stride, int M, int C, float eps,
                       int act, sgic_stream_t stream);

/* softmax(scale * Q K^T + bias) V for 64-wide heads; q/k/v/out are row-strided "token x (heads*64)"
 * matrices, head h at column h*64.  Sequence s, token t lives in row d_rowmap[s*L+t] (null: s*L+t) --
 * the Swin cyclic shift + window partition is such a map (blocks/swin_transformer.py:94-128).
 * d_bias: (nvar, L, L) additive (relative-position table + -inf shift masks), variant per sequence in
 * d_biasvar (null: 0).  nn.MultiheadAttention (titok/blocks.py:50-54) is the rowmap = bias = null case. */
int sgic_attention_f32(const float *d_q, int ldq, const float *d_k, int ldk, const float *d_v, int ldv,
                       float *d_out, int ldo, int L, int nseq, int nheads, const int32_t *d_rowmap,
                       const float *d_bias, const int32_t *d_biasvar, float scale, const sgic_launch_opts *opts,
                       sgic_stream_t stream);

/* im2col of non-overlapping PxP patches of an NCHW image with x*mul+add fused; patch rows in plain
 * (b,gy,gx) order or 16x16-tile-major (tile16) order (codec_sq_fixbpp.py:855,119; titok/blocks.py:98-100). */
int sgic_im2col_patch(const float *d_x, int B, int C, int H, int W, int P, float mul, float add, int tile16,
                      float *d_out, sgic_stream_t stream);
/* [cls+pos0 ; emb+pos ; lat+latpos] token assembly (codec_sq_fixbpp.py:127-138). */
int sgic_assemble_tokens(const float *d_emb, const float *d_cls, const float *d_pos, const float *d_lat,
                         const float *d_latpos, int N, int P, int T, int D, float *d_out, sgic_stream_t stream);
/* out[n*oseg+l,:] = in[n*iseg+l,:] + vec[l,:]  (models/cross_blocks.py:82-84); vec may be null. */
int sgic_add_rows_bcast(const float *d_in, int ldi, int iseg, const float *d_vec, float *d_out, int ldo, int oseg,
                        int Nn, int Lr, int D, sgic_stream_t stream);
/* depthwise kxk conv, NHWC, weights [k*k][C], optional per-channel prescale (blocks/conv_blocks.py:74-75,
 * blocks/dcvc.py:21,35). */
int sgic_dwconv_nhwc(const float *d_x, const float *d_w, const float *d_bias, const float *d_prescale, float *d_y,
                     int B, int H, int W, int C, int k, int tile16, sgic_stream_t stream);
/* im2col of the 2x2/s2 conv in feat_out (codec_sq_fixbpp.py:90): out[(b,y/2,x/2)][(ky*2+kx)*C+c]. */
int sgic_im2col_2x2(const float *d_x, int B, int H, int W, int C, int tile16, float *d_out, sgic_stream_t stream);
/* ConvFFN3 gate (blocks/dcvc.py:50-53). */
int sgic_gated_lrelu(const float *d_x, float *d_out, int M, int C2, sgic_stream_t stream);
/* y = x * v (mode 0), x / max(v, 0.5) (mode 1) or x * max(v, 0.5) (mode 2), v row m % vrows (sq_bottleneck.py:111,117;
 * compression_model.py:325-326,355). */
int sgic_colop(const float *d_x, int ldx, const float *d_v, int ldv, int vrows, float *d_y, int ldy, int M, int C,
               int mode, sgic_stream_t stream);
/* out[n][t][c] = in[n*in_seq_stride + c*T + t]  ("fake 2-D" reshape, codec_sq_fixbpp.py:175-177). */
int sgic_fake2d_transpose(const float *d_in, long in_seq_stride, float *d_out, int N, int T, int D,
                          sgic_stream_t stream);
/* l2-normalised nearest-code search (titok/quantizer.py:46-61). */
int sgic_vq_argmin(const float *d_z, int ldz, const float *d_codebook, int ncodes, int dim, int M, int l2norm,
                   int32_t *d_idx, sgic_stream_t stream);
/* CLIP preprocessing, bit-exact with ToPILImage -> PIL bicubic antialias resize -> center crop ->
 * ToTensor -> Normalize (compress.py:70-71).  Coefficient tables are Pillow's 22-bit fixed-point ints. */
int sgic_clip_preprocess(const float *d_x, long img_stride, long ch_stride, int ldx, int B, int H, int W, int OH,
                         int OW, int S, int top, int left, const int32_t *d_bounds_h, const int32_t *d_kk_h,
                         int ksize_h, const int32_t *d_bounds_v, const int32_t *d_kk_v, int ksize_v,
                         const float *mean3, const float *std3, uint8_t *d_tmp_u8, uint8_t *d_tmp_h, float *d_out,
                         sgic_stream_t stream);
/* unit-normalise rows + u8 quantise (compress.py:73,77). */
int sgic_l2norm_u8(const float *d_x, int ldx, int M, int D, float *d_unit, uint8_t *d_q, sgic_stream_t stream);

/* ---- fp32-accurate GEMM on the bf16 matrix pipe ("bf16x3" operand split; csrc/gemm_split.hip) -------------------
 * Same contract as sgic_gemm_f32 (the nn.Linear / 1x1 Conv2d call sites cited there), computed as six bf16 MFMAs per
 * 16 k over operands pre-split into three bf16 planes, x = x1 + x2 + x3 exactly.  Error vs an fp64 reference is at the
 * level of (measured: slightly below) a plain fp32 fmaf chain; results are bitwise independent of M, of the tile mode
 * and of the batch (fixed arithmetic order per output element), but NOT bitwise equal to sgic_gemm_f32's.
 * sgic_split3_f32: x[rows, cols] fp32 (row map as A of sgic_gemm_f32) -> d_planes [3][rows][cols] bf16 bit patterns;
 *   cols % 8 == 0.  Weights are split once at load time.
 * sgic_gemm_split3_f32: d_A != NULL: A is split into the caller's workspace d_Aplanes (3*M*K uint16) first;
 *   d_A == NULL: d_Aplanes already holds the planes.  K % 32 == 0.
 *   d_Cplanes != NULL: the result is written as slice-major planes [3][N / 32][M][32] (the next GEMM's A operand with
 *   opts->a_packed = 1) instead of d_C; N % 32 == 0.
 *   opts->tile_mode: 0 = heuristic, 1 = 128x256, 2 = 128x128, 3 = 64x64, 4 = 32x32 (the latency kernel for
 *   under-filled launches), 5 = 64x128 (two workgroups per CU) workgroup tiles, 6 / 7 = 1 / 2 for the rows that
 *   fill whole rounds of the 256 CUs + 5 for the remaining rows (two launches), 8 / 9 = the same with 4 for the remaining rows,
 *   10 / 11 = 1 / 2 as a persistent launch (one resident workgroup per CU walks the tile list), 12 / 13 = 8 / 9 with the
 *   whole rounds walked persistently, 14 / 15 = 256x128 tiles (plain / persistent), 16 / 17 = 2 / 11 with LDS-DMA staging,
 *   18-25 = the ring kernel's small tiles (32x32 ... 80x64: single-image sized launches), 26 / 27 = 8 / 12 with the ring kernel's
 *   80x64 tiles for the remaining rows, 28 / 29 = 128x192 tiles (plain / persistent; N = 768: 8192 rows = one whole round), 30 = 256x256 tiles
 *   (W single-buffered in LDS), 31 = 30 for the rows that fill whole rounds + 1 for the rest.
 *   The 128x256 and 256x128 tiles (1, 10, 14, 15 and the whole-round part of 6, 8, 12) stage their operands by LDS-DMA
 *   (global_load_lds: no register pass, no ds_write), the other tiles through registers; all modes are bitwise identical. */
int sgic_split3_f32(const float *d_x, int ld, int rows, int cols, int seg, int seg_stride, uint16_t *d_planes,
                    sgic_stream_t stream);
/* planes of an operand [rows][cols = K] in the slice-major layout [3][K / 32][rows][32] (opts->w_packed / a_packed = 1 at the
 * consuming call): a 16-row piece of a 32-k slice is 1 KiB of consecutive bytes (whole cache lines) for the LDS-DMA staging.  K % 32 == 0. */
int sgic_split3_pack_f32(const float *d_x, int ld, int rows, int cols, uint16_t *d_planes, sgic_stream_t stream);
int sgic_gemm_split3_f32(const float *d_A, int lda, int a_seg, int a_seg_stride, uint16_t *d_Aplanes,
                         const uint16_t *d_Wplanes, const float *d_bias, const float *d_R, int ldr, float *d_C, int ldc,
                         uint16_t *d_Cplanes, int M, int N, int K, int act, int c_seg, int c_seg_stride,
                         const sgic_launch_opts *opts, sgic_stream_t stream);
/* sgic_conv3x3_f32 as an implicit split GEMM: d_in_planes = bf16x3 planes of the zero-halo NHWC input, [3][B (H+2) (W+2)][Cin]
 * (sgic_split3_f32 over the halo buffer's rows; opts->a_packed = 0) or slice-major [3][Cin / 32][B (H+2) (W+2)][32]
 * (sgic_split3_pack_f32, sgic_groupnorm_nhwc_split3, sgic_halo_copy_split3; a_packed = 1); d_Wplanes = planes [3][Cout][9 Cin]
 * (or slice-major, w_packed = 1).  Cin % 32 == 0. */
int sgic_conv3x3_split3_f32(const uint16_t *d_in_planes, const uint16_t *d_Wplanes, const float *d_bias, const float *d_R,
                            int ldr, float *d_out, int ldc, int B, int H, int W, int Cin, int Cout, int act,
                            const sgic_launch_opts *opts, sgic_stream_t stream);
/* sgic_groupnorm_nhwc / sgic_halo_copy writing the interior of a zero-halo slice-major bf16x3 planes buffer
 * [3][C / 32][B (H+2) (W+2)][32] -- the input of sgic_conv3x3_split3_f32 with a_packed = 1 -- directly (no fp32 halo buffer, no
 * split pass).  C % 32 == 0. */
int sgic_groupnorm_nhwc_split3(const float *d_x, const float *d_gamma, const float *d_beta, int B, int H, int W, int C,
                               int groups, float eps, int swish, double *d_ws, float *d_stats, uint16_t *d_halo_planes,
                               sgic_stream_t stream);
int sgic_halo_copy_split3(const float *d_in, int B, int H, int W, int C, int upsample2x, int tile16, uint16_t *d_halo_planes,
                          sgic_stream_t stream);
/* sgic_attention_f32 with the output written as slice-major bf16x3 planes [3][nheads*2][rows][32] (rows = the row space of d_rowmap,
 * >= nseq*L): the A operand of the out-projection when that runs as a split GEMM. */
int sgic_attention_split3_f32(const float *d_q, int ldq, const float *d_k, int ldk, const float *d_v, int ldv,
                              uint16_t *d_out_planes, long rows, int L, int nseq, int nheads, const int32_t *d_rowmap,
                              const float *d_bias, const int32_t *d_biasvar, float scale, const sgic_launch_opts *opts,
                              sgic_stream_t stream);
/* nn.LayerNorm (call sites as sgic_layernorm_f32) whose only consumer is a split GEMM: the normalised rows are written
 * directly as slice-major bf16x3 planes [3][C / 32][M][32] (dense rows), so no fp32 copy and no separate split pass.  C % 256 == 0, C <= 2048. */
int sgic_layernorm_split3_f32(const float *d_x, int ldx, int xseg, int xseg_stride, const float *d_gamma,
                              const float *d_beta, uint16_t *d_planes, int M, int C, float eps, int act,
                              sgic_stream_t stream);

/* Batched GEMM (element strides, 0 = shared operand) -- VQGAN AttnBlock single-head attention as
 * S_b = Q_b K_b^T, O_b = P_b V_b (taming/modules/diffusionmodules/model.py:168-192). */
int sgic_gemm_batched_f32(const float *d_A, int lda, long strideA, const float *d_W, int ldw, long strideW,
                          const float *d_bias, const float *d_R, int ldr, long strideR, float *d_C, int ldc,
                          long strideC, int M, int N, int K, int act, int batch, const sgic_launch_opts *opts,
                          sgic_stream_t stream);
/* 3x3/s1/p1 Conv2d as an implicit GEMM on the matrix cores over a zero-halo NHWC input [B,H+2,W+2,Cin]
 * (Cin % 32 == 0); weights [Cout][(ky,kx,cin)]; fused bias/activation/residual (model.py:38-137,436-537). */
int sgic_conv3x3_f32(const float *d_in_halo, const float *d_W, const float *d_bias, const float *d_R, int ldr,
                     float *d_out, int ldc, int B, int H, int W, int Cin, int Cout, int act, const sgic_launch_opts *opts,
                     sgic_stream_t stream);
/* GroupNorm(groups, eps) on NHWC + optional swish; output plain or into the interior of a zero-halo buffer.
 * d_ws: B*64*C*2 doubles, d_stats: B*groups*2 floats (model.py:34-35,117-131). */
int sgic_groupnorm_nhwc(const float *d_x, const float *d_gamma, const float *d_beta, int B, int H, int W, int C,
                        int groups, float eps, int swish, int halo_out, double *d_ws, float *d_stats, float *d_y,
                        sgic_stream_t stream);
/* copy (optionally nearest-2x upsampled, optionally from tile-major rows) into the interior of a zero-halo
 * buffer [B, OH+2, OW+2, C] (Upsample, model.py:49-53). */
int sgic_halo_copy(const float *d_in, int B, int H, int W, int C, int upsample2x, int tile16, float *d_out,
                   sgic_stream_t stream);
/* y = softmax(scale * x) over rows of length L (model.py:181-183; codec_sq_fixbpp.py:660-661). */
int sgic_softmax_rows(const float *d_x, float *d_y, long M, int L, float scale, sgic_stream_t stream);
/* PixelShuffle(2) of a plain [(b,y,x), 4C] map into tile-major [(b,2y+i,2x+j), C] (codec_sq_fixbpp.py:203-207). */
int sgic_pixel_shuffle2_tm16(const float *d_in, int B, int H, int W, int C, float *d_out, sgic_stream_t stream);
/* decoder tokens [cls+pos0 ; mask+pos ; emb+latpos] (codec_sq_fixbpp.py:258-267). */
int sgic_assemble_dec_tokens(const float *d_emb, const float *d_cls, const float *d_mask, const float *d_pos,
                             const float *d_latpos, int N, int P, int T, int D, float *d_out, sgic_stream_t stream);
/* z_hat rows: codebook[idx] l2-normalised, zero-padded to ld floats (codec_sq_fixbpp.py:889-892). */
int sgic_codebook_gather_norm(const int32_t *d_idx, const float *d_codebook, int M, int dim, int ld, float *d_out,
                              sgic_stream_t stream);
/* x_hat [(b,y,x), ld>=3] -> clamp(-1,1) -> NCHW (codec_sq_fixbpp.py:901). */
int sgic_nhwc3_to_nchw_clamp(const float *d_in, int ld, int B, int H, int W, float *d_out, sgic_stream_t stream);

/* exact top-k per row, descending, ties -> lower index (IndexFlatIP.search, search.py:113-120); d_scores
 * (nq, n) = sgic_gemm_f32(queries, database) and is consumed (taken entries become -inf). */
int sgic_topk_rows(float *d_scores, int nq, int n, int k, float *d_out_scores, int32_t *d_out_idx, sgic_stream_t stream);

/* F.pad(x, (pl, pr, pt, pb), mode="replicate") on (BC, H, W) fp32 planes -> (BC, H+pt+pb, W+pl+pr)
 * (compress.py:258-261: every image is padded to a multiple of 256 before the encoder). */
int sgic_pad_replicate(const float *d_in, float *d_out, int BC, int H, int W, int pl, int pr, int pt, int pb,
                       sgic_stream_t stream);

/* Image ingest: decoded RGB u8 HWC (B, H, W, 3) on the device -> `transforms.ToTensor()(img) * 2.0 - 1.0` (compress.py:
 * 151-168) -> NCHW fp32 with F.pad(mode="replicate") fused (compress.py:258-261); out is (B, 3, H+pt+pb, W+pl+pr).
 * Bit-identical to the torch ops (one IEEE division, multiply, subtract per sample). */
int sgic_u8hwc_to_f32chw_pad(const uint8_t *d_in, float *d_out, int B, int H, int W, int pl, int pr, int pt, int pb,
                             sgic_stream_t stream);

/* Baseline JPEG decode of a batch of B equal-geometry files to RGB u8 HWC (B, H, W, 3) on the device -- the pixel decode inside
 * the reference's Test_Dataset (`Image.open(path).convert("RGB")`, compress.py:151-168), bit-exact with Pillow / libjpeg-turbo's
 * default decoder (islow IDCT, fancy chroma upsampling, jdcolor tables).  The host parses markers and strips byte stuffing
 * (sgic_amd/jpeg.py); Huffman decoding, IDCT, upsampling and colour conversion run here.  d_params: B x 64 int32 descriptors,
 * d_scan: cleaned entropy-coded segments (each padded to a multiple of 2048 B), d_tabs: B x 4 lookup tables of 1424 B,
 * d_segs: restart-interval byte offsets, d_quant: u16 tables in natural order -- these five may be device memory or PINNED host
 * memory (the kernels then pull the compressed bytes over PCIe themselves and no H2D copy exists); d_work_params (B x 64 int32),
 * d_work_quant (B x 256 u16), d_coef, d_planes: device workspaces; d_err[b]: 0 ok / 1 invalid code / 2 missing restart segment. */
int sgic_jpeg_decode_batch(const int32_t *d_params, const uint8_t *d_scan, const uint8_t *d_tabs, const int32_t *d_segs,
                           const uint16_t *d_quant, int32_t *d_work_params, uint16_t *d_work_quant, int16_t *d_coef,
                           uint8_t *d_planes, uint8_t *d_out, int32_t *d_err, int B, int H, int W, int max_blocks,
                           sgic_stream_t stream);

/* CLIP text tower front end: out[b*L+l,:] = table[ids[b,l],:] + pos[l,:] (open_clip CLIP.encode_text, reached from
 * search.py:93-97; ids outside [0,vocab) are clamped).  D multiple of 4. */
int sgic_embed_tokens(const int32_t *d_ids, const float *d_table, const float *d_pos, float *d_out, int B, int L, int D,
                      int vocab, sgic_stream_t stream);
/* CLIP text pooling: out[b,:] = x[b*L + argmax_l ids[b,l], :] (first maximum = the EOT token, the largest BPE id). */
int sgic_gather_eot_rows(const int32_t *d_ids, const float *d_x, int ldx, float *d_out, int B, int L, int D,
                         sgic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
