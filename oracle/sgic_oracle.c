/*
 * sgic_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the integer / byte arithmetic on the compress /
 * decompress hot path of lionl1106/Searchable-Generative-Image-Compression.  It is the *checker*
 * for the HIP path and the "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library (libsgic.so) never does.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_*.py) against
 *   - the real reference coder compiled from /root/reference/src/cpp into oracle/_ref (oracle/Makefile),
 *     through committed known-answer vectors in tests/golden/ (made by oracle/gen_golden.py), and
 *   - the reference's own worked example IO/bitstreams/apple.c2df (z-stream, container framing).
 *
 * Reference lines restated (paths relative to /root/reference/src):
 *   orc_pmf_to_quantized_cdf  cpp/ops/ops.cpp:24-82
 *   orc_rans_encode           cpp/rans/rans.cpp:101-159 (symbol mapping + bypass), :161-187 (flush),
 *                             cpp/rans/rans_byte.h:61-111, rans.cpp:35-51 (PutBits),
 *                             cpp/py_rans/py_rans.cpp:91-136 (1-byte stream flag, single stream)
 *   orc_rans_decode_*         cpp/rans/rans.cpp:280-362, rans_byte.h:115-155, rans.cpp:53-68,
 *                             cpp/py_rans/py_rans.cpp:150-185
 *   orc_pack12 / unpack12     models/codec_sq_fixbpp.py:841-846,863-864,886-887 (torchac 0.9.3 with a
 *                             uniform 4096-symbol CDF degenerates to 12-bit big-endian packing + 0x40)
 *   orc_quant_step            entropy/compression_model.py:224-239,296-366 (process_with_mask,
 *                             combine_for_writing), entropy/entropy_models.py:355-362,66-69
 *                             (build_indexes, clamp +-30000 -> int16)
 *   orc_resize_bicubic_u8     third-party Pillow ImagingResample (8bpc path) used by
 *                             compress.py:70-71 through open_clip's preprocess; restated from the
 *                             published algorithm, pinned against the PIL installed in the build image.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PRECISION 16
#define BYPASS_PRECISION 2
#define MAX_BYPASS_VAL 3
#define RANS_BYTE_L (1u << 23)

/* ------------------------------------------------------------------ */
/* pmf -> quantized cdf (ops.cpp:24-82)                                 */
/* ------------------------------------------------------------------ */
int orc_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf /* n+1 */) {
  cdf[0] = 0;
  for (int i = 0; i < n; i++) {
    /* static_cast<uint32_t>(std::round(p * (1 << precision)) + 0.5): float multiply, float round */
    float v = roundf(pmf[i] * (float)(1 << precision));
    cdf[i + 1] = (uint32_t)((double)v + 0.5);
  }
  /* std::accumulate(..., 0): int accumulator */
  int total_i = 0;
  for (int i = 0; i <= n; i++) total_i = (int)((uint32_t)total_i + cdf[i]);
  const uint32_t total = (uint32_t)total_i;
  if (total == 0) return -1;
  for (int i = 0; i <= n; i++) cdf[i] = (uint32_t)(((1ull << precision) * cdf[i]) / total);
  for (int i = 1; i <= n; i++) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;

  for (int i = 0; i < n; ++i) {
    if (cdf[i] == cdf[i + 1]) {
      uint32_t best_freq = ~0u;
      int best_steal = -1;
      for (int j = 0; j < n; ++j) {
        uint32_t freq = cdf[j + 1] - cdf[j];
        if (freq > 1 && freq < best_freq) {
          best_freq = freq;
          best_steal = j;
        }
      }
      if (best_steal == -1) return -2;
      if (best_steal < i) {
        for (int j = best_steal + 1; j <= i; ++j) cdf[j]--;
      } else {
        for (int j = i + 1; j <= best_steal; ++j) cdf[j]++;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* rANS encoder                                                        */
/* ------------------------------------------------------------------ */
typedef struct {
  uint16_t start, range;
} orc_sym_t;

/* Encode one stream: n (symbol,index) pairs (already the concatenation of all encode_with_indexes
 * calls between reset() and flush()).  cdf is [rows][cols] int32, sizes/offsets per row.
 * Output = 0x01 flag byte || rANS bytes (py_rans.cpp:91-136 with one encoder).
 * Returns number of bytes written, or <0 on error (-1: out of space, -2: bad index). */
long orc_rans_encode(const int16_t *sym, const int16_t *idx, size_t n, const int32_t *cdf, int rows,
                     int cols, const int32_t *sizes, const int32_t *offsets, uint8_t *out,
                     size_t cap) {
  /* pass 1: build the RansSymbol list exactly like rans.cpp:115-158 */
  size_t cap_syms = n * 14 + 16, ns = 0;
  orc_sym_t *syms = (orc_sym_t *)malloc(cap_syms * sizeof(orc_sym_t));
  if (!syms) return -1;
  for (size_t i = 0; i < n; ++i) {
    const int32_t cdf_idx = idx[i];
    if (cdf_idx < 0) continue;
    if (cdf_idx >= rows) {
      free(syms);
      return -2;
    }
    const int32_t max_value = sizes[cdf_idx] - 2;
    int32_t value = (int32_t)sym[i] - offsets[cdf_idx];
    uint32_t raw_val = 0;
    if (value < 0) {
      raw_val = (uint32_t)(-2 * value - 1);
      value = max_value;
    } else if (value >= max_value) {
      raw_val = (uint32_t)(2 * (value - max_value));
      value = max_value;
    }
    const int32_t *row = cdf + (size_t)cdf_idx * cols;
    syms[ns].start = (uint16_t)row[value];
    syms[ns].range = (uint16_t)(row[value + 1] - row[value]);
    ns++;
    if (value == max_value) {
      int32_t n_bypass = 0;
      while ((raw_val >> (n_bypass * BYPASS_PRECISION)) != 0) ++n_bypass;
      int32_t val = n_bypass;
      while (val >= MAX_BYPASS_VAL) {
        syms[ns].start = MAX_BYPASS_VAL;
        syms[ns].range = 0;
        ns++;
        val -= MAX_BYPASS_VAL;
      }
      syms[ns].start = (uint16_t)val;
      syms[ns].range = 0;
      ns++;
      for (int32_t j = 0; j < n_bypass; ++j) {
        syms[ns].start = (uint16_t)((raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL);
        syms[ns].range = 0;
        ns++;
      }
    }
  }
  /* pass 2: reverse walk (rans.cpp:161-187).  The reference sizes its scratch as ns bytes and
   * overruns on tiny/empty inputs; we use the true bound 2*ns+4. */
  size_t scratch_n = 2 * ns + 8;
  uint8_t *scratch = (uint8_t *)malloc(scratch_n);
  if (!scratch) {
    free(syms);
    return -1;
  }
  uint8_t *end = scratch + scratch_n, *ptr = end;
  uint32_t x = RANS_BYTE_L;
  for (size_t t = ns; t-- > 0;) {
    const orc_sym_t s = syms[t];
    if (s.range != 0) {
      const uint32_t freq = s.range, x_max = freq << 15;
      while (x >= x_max) {
        *(--ptr) = (uint8_t)(x & 0xff);
        x >>= 8;
      }
      x = ((x / freq) << PRECISION) + (x % freq) + s.start;
    } else {
      const uint32_t freq = 1u << (PRECISION - BYPASS_PRECISION), x_max = freq << 15;
      while (x >= x_max) {
        *(--ptr) = (uint8_t)(x & 0xff);
        x >>= 8;
      }
      x = (x << BYPASS_PRECISION) | s.start;
    }
  }
  ptr -= 4;
  ptr[0] = (uint8_t)(x >> 0);
  ptr[1] = (uint8_t)(x >> 8);
  ptr[2] = (uint8_t)(x >> 16);
  ptr[3] = (uint8_t)(x >> 24);
  size_t nbytes = (size_t)(end - ptr);
  long ret;
  if (nbytes + 1 > cap) {
    ret = -1;
  } else {
    out[0] = 0x01; /* ((1-1)<<4) + (perStreamHeader==2 ? 1 : 0) */
    memcpy(out + 1, ptr, nbytes);
    ret = (long)(nbytes + 1);
  }
  free(scratch);
  free(syms);
  return ret;
}

/* ------------------------------------------------------------------ */
/* rANS decoder with the reference's stateful cursor                   */
/* ------------------------------------------------------------------ */
typedef struct {
  const uint8_t *base;
  size_t len, pos;
  uint32_t x;
  int overrun;
} orc_dec_t;

static inline uint8_t orc_next(orc_dec_t *d) {
  if (d->pos < d->len) return d->base[d->pos++];
  d->overrun = 1; /* the reference reads past the end unchecked; we return 0 and flag it */
  return 0;
}

/* set_stream (py_rans.cpp:150-185 for one stream + rans.cpp:280-285).  Returns 0 / <0. */
int orc_rans_dec_init(orc_dec_t *d, const uint8_t *stream, size_t len) {
  memset(d, 0, sizeof(*d));
  if (len < 5) return -1;
  if ((stream[0] >> 4) != 0) return -3; /* multi-stream not used on this path */
  d->base = stream;
  d->len = len;
  d->pos = 1;
  uint32_t x = 0;
  x |= (uint32_t)orc_next(d) << 0;
  x |= (uint32_t)orc_next(d) << 8;
  x |= (uint32_t)orc_next(d) << 16;
  x |= (uint32_t)orc_next(d) << 24;
  d->x = x;
  return 0;
}

static inline uint32_t orc_get_bits(orc_dec_t *d, uint32_t nbits) {
  uint32_t x = d->x, val = x & ((1u << nbits) - 1);
  x >>= nbits;
  if (x < RANS_BYTE_L) x = (x << 8) | orc_next(d);
  d->x = x;
  return val;
}

/* decode_stream (rans.cpp:303-362).  Continues from the cursor left by the previous call. */
int orc_rans_decode(orc_dec_t *d, const int16_t *idx, size_t n, const int32_t *cdf, int rows, int cols,
                    const int32_t *sizes, const int32_t *offsets, int16_t *out) {
  for (size_t i = 0; i < n; ++i) {
    const int32_t cdf_idx = idx[i];
    if (cdf_idx < 0) {
      out[i] = 0;
      continue;
    }
    if (cdf_idx >= rows) return -2;
    const int32_t *row = cdf + (size_t)cdf_idx * cols;
    const int32_t size = sizes[cdf_idx], max_value = size - 2;
    const uint32_t cum = d->x & ((1u << PRECISION) - 1);
    int s = 0; /* first entry > cum, minus one (linear scan like std::find_if) */
    while (s < size && (uint32_t)row[s] <= cum) ++s;
    s -= 1;
    const uint32_t start = (uint32_t)row[s], freq = (uint32_t)(row[s + 1] - row[s]);
    uint32_t x = d->x;
    x = freq * (x >> PRECISION) + (x & ((1u << PRECISION) - 1)) - start;
    while (x < RANS_BYTE_L) x = (x << 8) | orc_next(d);
    d->x = x;
    int32_t value = s;
    if (value == max_value) {
      int32_t val = (int32_t)orc_get_bits(d, BYPASS_PRECISION);
      int32_t n_bypass = val;
      while (val == MAX_BYPASS_VAL) {
        val = (int32_t)orc_get_bits(d, BYPASS_PRECISION);
        n_bypass += val;
      }
      int32_t raw_val = 0;
      for (int j = 0; j < n_bypass; ++j) {
        val = (int32_t)orc_get_bits(d, BYPASS_PRECISION);
        raw_val |= val << (j * BYPASS_PRECISION);
      }
      value = raw_val >> 1;
      if (raw_val & 1)
        value = -value - 1;
      else
        value += max_value;
    }
    out[i] = (int16_t)(value + offsets[cdf_idx]);
  }
  return d->overrun ? -4 : 0;
}

/* one-shot helper: decode n symbols from a fresh stream */
int orc_rans_decode_all(const uint8_t *stream, size_t len, const int16_t *idx, size_t n, const int32_t *cdf,
                        int rows, int cols, const int32_t *sizes, const int32_t *offsets, int16_t *out) {
  orc_dec_t d;
  int rc = orc_rans_dec_init(&d, stream, len);
  if (rc) return rc;
  return orc_rans_decode(&d, idx, n, cdf, rows, cols, sizes, offsets, out);
}

size_t orc_dec_sizeof(void) { return sizeof(orc_dec_t); }

/* ------------------------------------------------------------------ */
/* z-branch stream: torchac(uniform 4096 cdf) == pack12 || 0x40        */
/* ------------------------------------------------------------------ */
size_t orc_pack12_size(size_t n) { return (n * 12 + 2 + 7) / 8; }

/* 12 bits per index, MSB first, followed by torchac's two terminating bits "01"
 * (pending_bits += 1; low < 2^30 -> emit 0 then the pending 1), zero-padded to a byte.
 * Even n: payload || 0x40.  Odd n: the "01" lands in the low nibble of the last payload byte. */
size_t orc_pack12(const int16_t *idx, size_t n, uint8_t *out) {
  size_t nb = orc_pack12_size(n);
  memset(out, 0, nb);
  for (size_t i = 0; i < n; i++) {
    uint32_t v = (uint32_t)idx[i] & 0xfff;
    size_t bit = i * 12;
    for (int b = 0; b < 12; b++)
      if (v & (1u << (11 - b))) out[(bit + b) >> 3] |= (uint8_t)(0x80 >> ((bit + b) & 7));
  }
  size_t tb = n * 12 + 1;
  out[tb >> 3] |= (uint8_t)(0x80 >> (tb & 7));
  return nb;
}

/* Full restatement of torchac 0.9.3's encoder (32-bit binary arithmetic coder, 16-bit cdf) for the
 * uniform cdf 16*i, kept so that odd token counts are also covered; for even counts it equals
 * orc_pack12.  Used to pin orc_pack12 and by the z-stream tests. */
typedef struct {
  uint8_t *out;
  size_t n, cap;
  uint8_t cache;
  int cached_bits;
} orc_bitw_t;
static void bw_put(orc_bitw_t *w, int bit) {
  w->cache = (uint8_t)((w->cache << 1) | (bit & 1));
  if (++w->cached_bits == 8) {
    if (w->n < w->cap) w->out[w->n] = w->cache;
    w->n++;
    w->cache = 0;
    w->cached_bits = 0;
  }
}
static void bw_put_pending(orc_bitw_t *w, int bit, uint64_t *pending) {
  bw_put(w, bit);
  while (*pending > 0) {
    bw_put(w, !bit);
    (*pending)--;
  }
}
size_t orc_torchac_uniform_encode(const int16_t *idx, size_t n, int lp /* = 4097 */, uint8_t *out,
                                  size_t cap) {
  orc_bitw_t w = {out, 0, cap, 0, 0};
  uint32_t low = 0, high = 0xFFFFFFFFu;
  uint64_t pending = 0;
  const int max_symbol = lp - 2;
  for (size_t i = 0; i < n; i++) {
    int s = idx[i];
    /* float cdf i/4096 -> int16 cdf: round(i/4096 * 65536) = 16*i (exact); last entry forced to 0 (=2^16) */
    uint32_t c_low = (uint32_t)(16 * s);
    uint32_t c_high = (s == max_symbol) ? 0x10000u : (uint32_t)(16 * (s + 1));
    uint64_t span = (uint64_t)high - (uint64_t)low + 1;
    high = (uint32_t)(low - 1 + ((span * c_high) >> 16));
    low = (uint32_t)(low + ((span * c_low) >> 16));
    for (;;) {
      if (high < 0x80000000u) {
        bw_put_pending(&w, 0, &pending);
        low <<= 1;
        high = (high << 1) | 1;
      } else if (low >= 0x80000000u) {
        bw_put_pending(&w, 1, &pending);
        low <<= 1;
        high = (high << 1) | 1;
      } else if (low >= 0x40000000u && high < 0xC0000000u) {
        pending++;
        low = (low << 1) & 0x7FFFFFFFu;
        high = (high << 1) | 0x80000001u;
      } else
        break;
    }
  }
  pending += 1;
  if (low < 0x40000000u)
    bw_put_pending(&w, 0, &pending);
  else
    bw_put_pending(&w, 1, &pending);
  if (w.cached_bits != 0) {
    uint8_t last = (uint8_t)(w.cache << (8 - w.cached_bits));
    if (w.n < w.cap) w.out[w.n] = last;
    w.n++;
  }
  return w.n;
}

void orc_unpack12(const uint8_t *in, size_t n, int16_t *idx) {
  for (size_t i = 0; i < n; i++) {
    size_t bit = i * 12;
    uint32_t v = 0;
    for (int b = 0; b < 12; b++) v = (v << 1) | ((in[(bit + b) >> 3] >> (7 - ((bit + b) & 7))) & 1);
    idx[i] = (int16_t)v;
  }
}

/* ------------------------------------------------------------------ */
/* one step of the 4-step masked quantiser + index builder (fp32)       */
/* ------------------------------------------------------------------ */
/* Layout here is the reference's NCHW: y, scales, means are (C=64,H,W) for ONE image, already
 * y = y / clamp_min(q_step, 0.5).  step k in 0..3.  Outputs (16,H,W) int16 symbols / indexes in the
 * reference's write order, and accumulates y_hat_so_far (C,H,W) += y_q + means*mask.
 * thr < 0 disables force_zero / skip. */
static inline float orc_rne(float v) { return nearbyintf(v); } /* torch.round = round half to even */

void orc_quant_step(const float *y, const float *scales, const float *means, int C, int H, int W, int k,
                    float thr, float *y_hat_so_far, int16_t *sym_out, int16_t *idx_out) {
  static const int xk[4] = {0, 3, 2, 1}; /* quarter q codes phase q ^ xk[k]  (compression_model.py:277-280) */
  const int Q = C / 4;
  const float log_min = (float)log(0.11);
  const float log_step = (float)((log(64.0) - log(0.11)) / 255.0);
  for (int c = 0; c < Q; c++)
    for (int i = 0; i < H; i++)
      for (int j = 0; j < W; j++) {
        const int p = (i & 1) * 2 + (j & 1);
        const int q = p ^ xk[k];
        const size_t a = ((size_t)(q * Q + c) * H + i) * W + j;
        const float mu = means[a], sg = scales[a];
        float r = y[a] - mu;
        float s = orc_rne(r);
        float sg_hat = sg;
        if (thr >= 0.f && sg < thr) {
          s = 0.f;
          sg_hat = 0.f;
        }
        y_hat_so_far[a] += s + mu;
        /* build_indexes (entropy_models.py:355-362) */
        float sc = fmaxf(sg_hat, 1e-5f);
        float fi = ((float)log((double)sc) - log_min) / log_step; /* correctly-rounded logf, then fp32 */
        fi = fminf(fmaxf(fi, 0.f), 255.f);
        int ii = (int)fi;
        if (thr >= 0.f && sg_hat < thr) ii = -1;
        float sc2 = fminf(fmaxf(s, -30000.f), 30000.f);
        const size_t o = ((size_t)c * H + i) * W + j;
        sym_out[o] = (int16_t)sc2;
        idx_out[o] = (int16_t)ii;
      }
}

/* ------------------------------------------------------------------ */
/* Pillow 8bpc bicubic (antialias) resize, RGB planar u8               */
/* ------------------------------------------------------------------ */
#define PIL_PRECISION_BITS (32 - 8 - 2)
static double orc_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static uint8_t orc_clip8(int v) {
  v >>= PIL_PRECISION_BITS;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}
/* returns ksize; bounds[2*out] (xmin, count); kk[out*ksize] fixed-point ints */
int orc_pil_coeffs(int in_size, int out_size, int *bounds, int *kk, int kk_cap) {
  double scale = (double)in_size / (double)out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  if (out_size * ksize > kk_cap) return -1;
  double *k = (double *)malloc(sizeof(double) * ksize);
  for (int xx = 0; xx < out_size; xx++) {
    double center = (xx + 0.5) * scale, ww = 0.0, ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x;
    for (x = 0; x < xmax; x++) {
      double w = orc_bicubic((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; x++)
      if (ww != 0.0) k[x] /= ww;
    for (; x < ksize; x++) k[x] = 0;
    for (x = 0; x < ksize; x++) {
      if (k[x] < 0)
        kk[xx * ksize + x] = (int)(-0.5 + k[x] * (1 << PIL_PRECISION_BITS));
      else
        kk[xx * ksize + x] = (int)(0.5 + k[x] * (1 << PIL_PRECISION_BITS));
    }
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  free(k);
  return ksize;
}

/* in: (3,H,W) u8 planar -> out: (3,OH,OW); horizontal pass then vertical pass like ImagingResample */
int orc_resize_bicubic_u8(const uint8_t *in, int H, int W, uint8_t *out, int OH, int OW) {
  int *bh = (int *)malloc(sizeof(int) * 2 * OW), *bv = (int *)malloc(sizeof(int) * 2 * OH);
  int capw = OW * ((int)ceil(2.0 * fmax(1.0, (double)W / OW)) * 2 + 1);
  int caph = OH * ((int)ceil(2.0 * fmax(1.0, (double)H / OH)) * 2 + 1);
  int *kh = (int *)malloc(sizeof(int) * capw), *kv = (int *)malloc(sizeof(int) * caph);
  int ksh = orc_pil_coeffs(W, OW, bh, kh, capw), ksv = orc_pil_coeffs(H, OH, bv, kv, caph);
  uint8_t *tmp = (uint8_t *)malloc((size_t)3 * H * OW);
  const int need_h = (OW != W), need_v = (OH != H);
  for (int c = 0; c < 3; c++) {
    for (int y = 0; y < H; y++)
      for (int xx = 0; xx < OW; xx++) {
        if (!need_h) {
          tmp[((size_t)c * H + y) * OW + xx] = in[((size_t)c * H + y) * W + xx];
          continue;
        }
        int xmin = bh[2 * xx], cnt = bh[2 * xx + 1], ss = 1 << (PIL_PRECISION_BITS - 1);
        for (int x = 0; x < cnt; x++) ss += in[((size_t)c * H + y) * W + xmin + x] * kh[xx * ksh + x];
        tmp[((size_t)c * H + y) * OW + xx] = orc_clip8(ss);
      }
    for (int yy = 0; yy < OH; yy++)
      for (int xx = 0; xx < OW; xx++) {
        if (!need_v) {
          out[((size_t)c * OH + yy) * OW + xx] = tmp[((size_t)c * H + yy) * OW + xx];
          continue;
        }
        int ymin = bv[2 * yy], cnt = bv[2 * yy + 1], ss = 1 << (PIL_PRECISION_BITS - 1);
        for (int y = 0; y < cnt; y++) ss += tmp[((size_t)c * H + ymin + y) * OW + xx] * kv[yy * ksv + y];
        out[((size_t)c * OH + yy) * OW + xx] = orc_clip8(ss);
      }
  }
  free(tmp);
  free(kh);
  free(kv);
  free(bh);
  free(bv);
  return 0;
}
