/*
 * sgic.h -- C ABI of libsgic.so, the MI355X (gfx950) hot path of the searchable generative image codec.
 *
 * Every entry point is `extern "C"`, takes plain pointers + sizes (no torch / pybind types), returns
 * 0 on success or a negative SGIC_E* code, never throws, and launches on the HIP stream it is given
 * (pass NULL for the default stream).  Pointers named d_* are DEVICE pointers (HBM); everything else is
 * host memory.  Handles are thread-compatible: one handle per thread/stream.
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
/* Decoder twins (compression_model.py:377-418): indexes from scales for step k, and
 * y_hat[active] = sym + mean after the symbols were decoded. */
int sgic_index_step(const float *d_scales, int ld_sm, int B, int H, int W, int C, int k, float thr,
                    int16_t *d_idx, sgic_stream_t stream);
int sgic_dequant_step(const int16_t *d_sym, const float *d_means, int ld_sm, float *d_yhat, int ld_yhat, int B,
                      int H, int W, int C, int k, sgic_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
