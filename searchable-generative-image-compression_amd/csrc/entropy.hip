// Entropy-coder kernels for gfx950: batched rANS encode/decode (one workgroup per image stream),
// 12-bit z-stream packing, the fused 4-step masked quantiser + index builder, and the host-side
// CDF table builder.  Replaces the reference's C++ coder (src/cpp/rans, src/cpp/py_rans, src/cpp/ops)
// and the torch elementwise chain around it (entropy/compression_model.py:224-239,296-366,
// entropy/entropy_models.py:355-374).  Integer results are bit-exact with the reference.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include <limits.h>

namespace sgic {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace sgic

extern "C" const char *sgic_last_error(void) { return sgic::g_err; }
extern "C" int sgic_version(void) { return 100; }
extern "C" int sgic_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SGIC_ENODEV;
  return n;
}

// ------------------------------------------------------------------------------------------------
// pmf -> quantized cdf  (ops.cpp:24-82); host, once per process
// ------------------------------------------------------------------------------------------------
extern "C" int sgic_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
  SGIC_REQUIRE(pmf && cdf && n > 0 && precision > 0 && precision <= 16, "args");
  cdf[0] = 0;
  for (int i = 0; i < n; i++) cdf[i + 1] = (uint32_t)((double)roundf(pmf[i] * (float)(1 << precision)) + 0.5);
  uint32_t total = 0;
  for (int i = 0; i <= n; i++) total += cdf[i];
  SGIC_REQUIRE(total != 0, "all-zero pmf");
  for (int i = 0; i <= n; i++) cdf[i] = (uint32_t)(((1ull << precision) * cdf[i]) / total);
  for (int i = 1; i <= n; i++) cdf[i] += cdf[i - 1];
  cdf[n] = 1u << precision;
  for (int i = 0; i < n; ++i) {
    if (cdf[i] != cdf[i + 1]) continue;
    uint32_t best = ~0u;
    int steal = -1;
    for (int j = 0; j < n; ++j) {
      uint32_t f = cdf[j + 1] - cdf[j];
      if (f > 1 && f < best) best = f, steal = j;
    }
    SGIC_REQUIRE(steal != -1, "no frequency left to steal");
    if (steal < i)
      for (int j = steal + 1; j <= i; ++j) cdf[j]--;
    else
      for (int j = i + 1; j <= steal; ++j) cdf[j]++;
  }
  return SGIC_OK;
}

// ------------------------------------------------------------------------------------------------
// CDF table in HBM.  enc[r*cols + v] = start | freq<<16 (freq of 65536 cannot occur: >= 2 bins).
// ------------------------------------------------------------------------------------------------
struct sgic_cdf_table {
  int rows, cols;
  uint32_t *d_enc;    // rows*cols  start | freq << 16
  int32_t *d_cdf;     // rows*cols  cdf, INT_MAX past each row's length (decoder probe)
  int32_t *d_sizes;   // rows
  int32_t *d_offsets; // rows
  size_t dec_lds_set; // dynamic-LDS limit already raised for rans_decode_kernel on this table's device (idempotent)
};

extern "C" int sgic_cdf_table_create(const int32_t *cdf, int rows, int cols, const int32_t *sizes,
                                     const int32_t *offsets, sgic_cdf_table **out) {
  SGIC_REQUIRE(cdf && sizes && offsets && out && rows > 0 && cols >= 3, "args");
  std::vector<uint32_t> enc((size_t)rows * cols, 0);
  for (int r = 0; r < rows; r++) {
    SGIC_REQUIRE(sizes[r] >= 3 && sizes[r] <= cols, "cdf row size");
    for (int v = 0; v + 1 < sizes[r]; v++) {
      int32_t s = cdf[(size_t)r * cols + v], f = cdf[(size_t)r * cols + v + 1] - s;
      SGIC_REQUIRE(s >= 0 && f > 0 && s + f <= 65536 && f < 65536, "cdf not strictly increasing");
      enc[(size_t)r * cols + v] = (uint32_t)s | ((uint32_t)f << 16);
    }
  }
  sgic_cdf_table *t = new sgic_cdf_table();
  t->rows = rows;
  t->cols = cols;
  size_t nb = (size_t)rows * cols * 4;
  SGIC_HIP(hipMalloc(&t->d_enc, nb));
  SGIC_HIP(hipMalloc(&t->d_cdf, nb));
  SGIC_HIP(hipMalloc(&t->d_sizes, rows * 4));
  SGIC_HIP(hipMalloc(&t->d_offsets, rows * 4));
  SGIC_HIP(hipMemcpy(t->d_enc, enc.data(), nb, hipMemcpyHostToDevice));
  // decoder image of the table: entries past a row's length are INT_MAX, so the lane-parallel probe "row[j] <= cum"
  // needs no row length (it is also false for the final entry 65536, cum being 16 bits)
  std::vector<int32_t> padded(cdf, cdf + (size_t)rows * cols);
  for (int r = 0; r < rows; r++)
    for (int v = sizes[r]; v < cols; v++) padded[(size_t)r * cols + v] = INT_MAX;
  SGIC_HIP(hipMemcpy(t->d_cdf, padded.data(), nb, hipMemcpyHostToDevice));
  SGIC_HIP(hipMemcpy(t->d_sizes, sizes, rows * 4, hipMemcpyHostToDevice));
  SGIC_HIP(hipMemcpy(t->d_offsets, offsets, rows * 4, hipMemcpyHostToDevice));
  *out = t;
  return SGIC_OK;
}

extern "C" void sgic_cdf_table_destroy(sgic_cdf_table *t) {
  if (!t) return;
  (void)hipFree(t->d_enc);
  (void)hipFree(t->d_cdf);
  (void)hipFree(t->d_sizes);
  (void)hipFree(t->d_offsets);
  delete t;
}

// ------------------------------------------------------------------------------------------------
// rANS encode: one 256-thread workgroup per image.
//   phase A (all threads, per 1024-symbol chunk walking from the END of the image's list): table
//            lookup -> {start|freq, exact reciprocal of freq, bypass raw value} into LDS;
//   phase B (thread 0): the inherently serial state recurrence x -> C(s,x), bytes stored backwards
//            into the image's output slot, end-aligned (no compaction pass).
// The reciprocal (Alverson, as in ryg_rans' RansEncSymbolInit) makes x/freq a mul_hi+shift and is exact
// for x < 2^31, so the bytes equal the reference's ((x/freq)<<16)+(x%freq)+start.
// ------------------------------------------------------------------------------------------------
#define ENC_CHUNK 1024
#define RANS_L (1u << 23)
#define RAW_SKIP 0xFFFFFFFFu
#define RAW_NONE 0xFFFFFFFEu

__global__ __launch_bounds__(256) void rans_encode_kernel(const int16_t *__restrict__ sym,
                                                          const int16_t *__restrict__ idx, int n,
                                                          const uint32_t *__restrict__ enc,
                                                          const int32_t *__restrict__ sizes,
                                                          const int32_t *__restrict__ offsets, int rows, int cols,
                                                          uint8_t *__restrict__ out, int cap, int32_t *d_off,
                                                          int32_t *d_len, int32_t *d_err) {
  // one 16-byte record per symbol: {start | freq << 16, reciprocal, RAW_SKIP / RAW_NONE / bypass raw value, shift}
  __shared__ __attribute__((aligned(16))) uint4 s_rec[ENC_CHUNK];
  __shared__ int s_bad;
  const int b = blockIdx.x, tid = threadIdx.x;
  sym += (size_t)b * n;
  idx += (size_t)b * n;
  uint8_t *slot = out + (size_t)b * cap;
  if (tid == 0) s_bad = 0;
  uint32_t x = RANS_L;
  int ptr = cap;  // bytes slot[ptr..cap) are written; carried (uniformly) by every lane of wave 0
  int err = 0;
  __syncthreads();

  for (int hi = n; hi > 0; hi -= ENC_CHUNK) {
    const int lo = hi > ENC_CHUNK ? hi - ENC_CHUNK : 0;
    const int cnt = hi - lo;
    for (int t = tid; t < cnt; t += 256) {
      const int ci = idx[lo + t];
      uint32_t sf = 0, rcp = 0, raw = RAW_SKIP, sh = 0;
      if (ci >= 0) {
        if (ci >= rows) {
          s_bad = 1;
        } else {
          const int max_value = sizes[ci] - 2;
          int value = (int)sym[lo + t] - offsets[ci];
          raw = RAW_NONE;
          if (value < 0) {
            raw = (uint32_t)(-2 * value - 1);
            value = max_value;
          } else if (value >= max_value) {
            raw = (uint32_t)(2 * (value - max_value));
            value = max_value;
          }
          sf = enc[(size_t)ci * cols + value];
          const uint32_t freq = sf >> 16;
          if (freq < 2) {
            rcp = ~0u;
            sh = 0;
          } else {
            uint32_t shift = 0;
            while (freq > (1u << shift)) shift++;
            rcp = (uint32_t)(((1ull << (shift + 31)) + freq - 1) / freq);
            sh = shift - 1;
          }
        }
      }
      s_rec[t] = make_uint4(sf, rcp, raw, sh);
    }
    __syncthreads();
    // Phase B on the SCALAR unit: wave 0 runs the recurrence with uniform control flow -- every lane carries the same
    // (x, ptr, err), the record of the NEXT symbol is requested while the current one is folded in (one broadcast
    // ds_read_b128 per symbol instead of four dependent reads), its fields are moved to SGPRs (readfirstlane) so the state
    // update is s_mul_hi / s_lshr / s_add, and only lane 0 stores the bytes.  (As a single lane on the vector unit this
    // loop cost ~500 cycles per symbol: 0.9 ms for one image, serial on the critical path of a single-image request.)
    if (tid < 64 && !err && !s_bad) {
      const bool l0 = tid == 0;
      // scalar copies: readfirstlane tells the compiler the state is wave-uniform, so the recurrence compiles to SALU
      uint32_t xs = __builtin_amdgcn_readfirstlane(x);
      int ps = __builtin_amdgcn_readfirstlane(ptr);
      uint4 rec_n = s_rec[cnt - 1];
      for (int t = cnt - 1; t >= 0; --t) {
        const uint4 rec = rec_n;
        rec_n = s_rec[t > 0 ? t - 1 : 0];
        const uint32_t raw = __builtin_amdgcn_readfirstlane(rec.z);
        if (raw == RAW_SKIP) continue;
        if (ps < 48) {  // worst case for one symbol: 11 bypass entries + 1 symbol < 16 bytes (+ flush/flag)
          err = SGIC_ENOSPC;
          break;
        }
        if (raw != RAW_NONE) {
          // reversed emission of: [n_bypass in base-3 "unary"], then n_bypass 2-bit digits LSB first
          int nb = 0;
          while ((raw >> (2 * nb)) != 0) ++nb;
          for (int j = nb - 1; j >= 0; --j) {
            const uint32_t v = (raw >> (2 * j)) & 3u;
            while (xs >= (1u << 29)) {
              --ps;
              if (l0) slot[ps] = (uint8_t)xs;
              xs >>= 8;
            }
            xs = (xs << 2) | v;
          }
          const int n3 = nb / 3, rem = nb - 3 * n3;
          for (int j = 0; j <= n3; ++j) {
            const uint32_t v = (j == 0) ? (uint32_t)rem : 3u;
            while (xs >= (1u << 29)) {
              --ps;
              if (l0) slot[ps] = (uint8_t)xs;
              xs >>= 8;
            }
            xs = (xs << 2) | v;
          }
        }
        const uint32_t sf = __builtin_amdgcn_readfirstlane(rec.x), freq = sf >> 16, start = sf & 0xffffu;
        const uint32_t x_max = freq << 15;
        while (xs >= x_max) {
          --ps;
          if (l0) slot[ps] = (uint8_t)xs;
          xs >>= 8;
        }
        const uint32_t q = __umulhi(xs, __builtin_amdgcn_readfirstlane(rec.y)) >> __builtin_amdgcn_readfirstlane(rec.w);
        const uint32_t bias = (freq < 2) ? start + 65535u : start;
        xs = xs + bias + q * (65536u - freq);
      }
      x = xs;
      ptr = ps;
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (s_bad) err = SGIC_EINVAL;
    if (!err) {
      ptr -= 4;
      slot[ptr + 0] = (uint8_t)(x >> 0);
      slot[ptr + 1] = (uint8_t)(x >> 8);
      slot[ptr + 2] = (uint8_t)(x >> 16);
      slot[ptr + 3] = (uint8_t)(x >> 24);
      slot[--ptr] = 0x01;  // single-stream flag (py_rans.cpp:118-120)
    }
    d_off[b] = err ? cap : ptr;
    d_len[b] = err ? 0 : cap - ptr;
    d_err[b] = err;
  }
}

extern "C" int sgic_rans_encode_batch(const sgic_cdf_table *t, const int16_t *d_sym, const int16_t *d_idx, int B,
                                      int n_per_img, uint8_t *d_out, int cap, int32_t *d_off, int32_t *d_len,
                                      int32_t *d_err, sgic_stream_t stream) {
  SGIC_REQUIRE(t && d_out && d_off && d_len && d_err && B > 0 && n_per_img >= 0 && cap >= 64, "args");
  SGIC_REQUIRE(n_per_img == 0 || (d_sym && d_idx), "null symbols");
  rans_encode_kernel<<<B, 256, 0, to_stream(stream)>>>(d_sym, d_idx, n_per_img, t->d_enc, t->d_sizes, t->d_offsets,
                                                        t->rows, t->cols, d_out, cap, d_off, d_len, d_err);
  return sgic::check_launch("rans_encode_kernel");
}

// ------------------------------------------------------------------------------------------------
// rANS decode: serial per stream, one lane per image (the 4-step prior makes decode a true
// dependency chain: NN -> indexes -> decode -> NN ...).  Cursor {x,pos,err} lives in HBM between calls.
// ------------------------------------------------------------------------------------------------
struct DecCursor {
  const uint8_t *p;
  int pos, len;
  int err;
  __device__ inline uint32_t next() {
    if (pos < len) return p[pos++];
    err = 1;  // the reference over-reads silently (rans.cpp:53-68); we flag it
    return 0;
  }
};

__global__ void rans_decode_init_kernel(const uint8_t *streams, int cap, const int32_t *d_off, const int32_t *d_len,
                                        int B, uint32_t *state) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int off = d_off ? d_off[b] : 0;
  DecCursor c{streams + (size_t)b * cap + off, 0, d_len[b], 0};
  uint32_t flag = c.next();
  if ((flag >> 4) != 0) c.err = 2;  // multi-stream containers are not produced by this codec
  uint32_t x = c.next();
  x |= c.next() << 8;
  x |= c.next() << 16;
  x |= c.next() << 24;
  state[b * 4 + 0] = x;
  state[b * 4 + 1] = (uint32_t)c.pos;
  state[b * 4 + 2] = (uint32_t)c.err;
  state[b * 4 + 3] = 0;
}

// One workgroup (256 threads) per image.  The whole quantised-cdf table (rows x cols int32, 105 KB for the codec's
// 256 x 103 table) and a window of the stream are staged in LDS; wave 0 then walks the symbols.  The recurrence is
// serial, but the cdf search is not: lane j compares row[j] and row[j+64] with the 16-bit cumulative value and one
// ballot + popcount gives the symbol (rows are strictly increasing), instead of 7 dependent binary-search probes.
// Symbol indexes are fetched and results stored 64 at a time, coalesced.
#define DEC_WIN 4096
__global__ __launch_bounds__(256) void rans_decode_kernel(const int32_t *__restrict__ cdf, const int32_t *__restrict__ sizes,
                                                          const int32_t *__restrict__ offsets, int rows, int cols,
                                                          const uint8_t *__restrict__ streams, int cap, const int32_t *d_off,
                                                          const int32_t *d_len, int B, uint32_t *state,
                                                          const int16_t *__restrict__ idx, int n, int idx_stride,
                                                          int16_t *__restrict__ out, int out_stride) {
  extern __shared__ __attribute__((aligned(16))) int32_t s_dyn[];
  int32_t *s_cdf = s_dyn;                                   // [rows * cols]
  int32_t *s_size = s_dyn + rows * cols;                    // [rows]
  int32_t *s_offs = s_size + rows;                          // [rows]
  uint8_t *s_win = reinterpret_cast<uint8_t *>(s_offs + rows);   // [DEC_WIN] stream bytes from the cursor on
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int off = d_off ? d_off[b] : 0;
  const uint8_t *sp = streams + (size_t)b * cap + off;
  const int len = d_len[b], pos0 = (int)state[b * 4 + 1];
  {  // the table image is already padded with INT_MAX past each row's length (sgic_cdf_table_create)
    const int total = rows * cols, n4 = ((reinterpret_cast<uintptr_t>(cdf) & 15) == 0) ? total >> 2 : 0;
    for (int i = tid; i < n4; i += 256) reinterpret_cast<int4 *>(s_cdf)[i] = reinterpret_cast<const int4 *>(cdf)[i];
    for (int i = n4 * 4 + tid; i < total; i += 256) s_cdf[i] = cdf[i];
  }
  for (int i = tid; i < rows; i += 256) s_size[i] = sizes[i], s_offs[i] = offsets[i];
  for (int i = tid; i < DEC_WIN; i += 256) s_win[i] = (pos0 + i < len) ? sp[pos0 + i] : (uint8_t)0;
  __syncthreads();
  if (tid >= 64) return;

  // every lane of wave 0 carries the same cursor (uniform control flow); the cdf row and the next 64 stream bytes
  // live one element per lane and are picked with v_readlane, so the serial chain per symbol is
  // probe (prefetched) -> ballot -> readlane x2 -> multiply-add -> renormalise.
  uint32_t x = state[b * 4 + 0];
  int pos = pos0, err = (int)state[b * 4 + 2];
  int wb = 0;  // base of the 64-byte block held in my_byte, relative to pos0
  auto load_block = [&]() -> uint32_t {
    const int w = wb + lane;
    return (w < DEC_WIN) ? (uint32_t)s_win[w] : ((pos0 + w < len) ? (uint32_t)sp[pos0 + w] : 0u);
  };
  uint32_t my_byte = load_block();
  auto next = [&]() -> uint32_t {
    if (pos >= len) {
      err = 1;  // the reference over-reads silently (rans.cpp:53-68); we flag it
      return 0;
    }
    const int w = pos - pos0;
    if (w - wb >= 64) {
      wb += 64;
      my_byte = load_block();
    }
    ++pos;
    return (uint32_t)__builtin_amdgcn_readlane((int)my_byte, __builtin_amdgcn_readfirstlane(w - wb));
  };
  auto probe = [&](int ci, int32_t &v0, int32_t &v1, int &sz) {
    const int cc = (ci >= 0 && ci < rows) ? ci : 0;
    const int32_t *row = s_cdf + cc * cols;
    v0 = lane < cols ? row[lane] : INT_MAX;
    v1 = lane + 64 < cols ? row[lane + 64] : INT_MAX;
    sz = s_size[cc];
  };
  idx += (size_t)b * idx_stride;
  out += (size_t)b * out_stride;
  for (int base = 0; base < n; base += 64) {
    const int cnt = min(64, n - base);
    const int my_ci = (lane < cnt) ? (int)idx[base + lane] : -1;
    int my_val = 0;
    int ci_n = __builtin_amdgcn_readlane(my_ci, 0);
    int32_t v0n, v1n;
    int szn;
    probe(ci_n, v0n, v1n, szn);
    for (int k = 0; k < cnt; ++k) {
      const int ci = ci_n;
      const int32_t v0 = v0n, v1 = v1n;
      const int max_value = szn - 2;
      if (k + 1 < cnt) {  // next symbol's row is fetched while this one is decoded
        ci_n = __builtin_amdgcn_readlane(my_ci, __builtin_amdgcn_readfirstlane(k + 1));
        probe(ci_n, v0n, v1n, szn);
      }
      int value = 0;
      if (ci >= 0) {
        if (ci >= rows || err) {
          err = err ? err : 3;
        } else {
          const uint32_t cum = x & 0xffffu;
          // number of j with row[j] <= cum  ==  (largest such j) + 1 ; row[0] = 0 so it is >= 1
          const int lo = __builtin_amdgcn_readfirstlane(__popcll(__ballot(v0 <= (int32_t)cum)) + __popcll(__ballot(v1 <= (int32_t)cum)) - 1);
          const int hi = lo + 1;
          const uint32_t start = (uint32_t)(lo < 64 ? __builtin_amdgcn_readlane(v0, lo & 63) : __builtin_amdgcn_readlane(v1, lo & 63));
          const uint32_t end = (uint32_t)(hi < 64 ? __builtin_amdgcn_readlane(v0, hi & 63) : __builtin_amdgcn_readlane(v1, hi & 63));
          x = (end - start) * (x >> 16) + cum - start;
          while (x < RANS_L) x = (x << 8) | next();
          value = lo;
          if (value == max_value) {  // the last bin of a row is the bypass sentinel
            auto bits2 = [&]() {
              uint32_t v = x & 3u;
              x >>= 2;
              if (x < RANS_L) x = (x << 8) | next();
              return (int)v;
            };
            int val = bits2(), nb = val;
            while (val == 3 && !err) {
              val = bits2();
              nb += val;
            }
            int raw = 0;
            for (int j = 0; j < nb && j < 16; ++j) raw |= bits2() << (2 * j);
            value = raw >> 1;
            value = (raw & 1) ? -value - 1 : value + max_value;
          }
          value += s_offs[ci];
        }
      }
      if (lane == k) my_val = value;
    }
    if (lane < cnt) out[base + lane] = (int16_t)my_val;
  }
  if (lane == 0) {
    state[b * 4 + 0] = x;
    state[b * 4 + 1] = (uint32_t)pos;
    state[b * 4 + 2] = (uint32_t)err;
  }
}

extern "C" int sgic_rans_decode_init_batch(const uint8_t *d_streams, int cap, const int32_t *d_off,
                                           const int32_t *d_len, int B, uint32_t *d_state, sgic_stream_t stream) {
  SGIC_REQUIRE(d_streams && d_len && d_state && B > 0 && cap > 0, "args");
  rans_decode_init_kernel<<<cdiv(B, 64), 64, 0, to_stream(stream)>>>(d_streams, cap, d_off, d_len, B, d_state);
  return sgic::check_launch("rans_decode_init_kernel");
}

extern "C" int sgic_rans_decode_batch(const sgic_cdf_table *t, const uint8_t *d_streams, int cap,
                                      const int32_t *d_off, const int32_t *d_len, int B, uint32_t *d_state,
                                      const int16_t *d_idx, int n, int idx_stride, int16_t *d_sym_out,
                                      int out_stride, sgic_stream_t stream) {
  SGIC_REQUIRE(t && d_streams && d_len && d_state && d_idx && d_sym_out && B > 0 && n >= 0, "args");
  SGIC_REQUIRE(idx_stride >= n && out_stride >= n, "strides");
  SGIC_REQUIRE(t->cols <= 128, "the lane-parallel cdf probe covers rows of at most 128 entries");
  // one workgroup per image: table + stream window in LDS (dynamic: 105 KB for the 256 x 103 table)
  const size_t lds = ((size_t)t->rows * t->cols + 2 * (size_t)t->rows) * sizeof(int32_t) + DEC_WIN;
  SGIC_REQUIRE(lds <= 160 * 1024, "cdf table does not fit the 160 KB LDS");
  // the attribute is per device; the table lives on one device, so the "already raised" note is kept in the table (no
  // process-global state; two threads racing here both set the same value)
  sgic_cdf_table *tm = const_cast<sgic_cdf_table *>(t);
  if (lds > tm->dec_lds_set) {
    SGIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(rans_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    tm->dec_lds_set = lds;
  }
  rans_decode_kernel<<<B, 256, lds, to_stream(stream)>>>(t->d_cdf, t->d_sizes, t->d_offsets, t->rows, t->cols, d_streams,
                                                          cap, d_off, d_len, B, d_state, d_idx, n, idx_stride, d_sym_out,
                                                          out_stride);
  return sgic::check_launch("rans_decode_kernel");
}

// ------------------------------------------------------------------------------------------------
// 12-bit packing of the TiTok indices (== torchac with the uniform cdf)
// ------------------------------------------------------------------------------------------------
extern "C" size_t sgic_pack12_size(size_t n) { return (n * 12 + 2 + 7) / 8; }

__global__ void pack12_kernel(const int32_t *__restrict__ idx, int n, uint8_t *__restrict__ out, int nbytes) {
  const int b = blockIdx.y;
  idx += (size_t)b * n;
  out += (size_t)b * nbytes;
  for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < nbytes; o += gridDim.x * blockDim.x) {
    // byte o covers bits [8o, 8o+8)
    uint32_t v = 0;
    for (int bit = 0; bit < 8; ++bit) {
      const int g = o * 8 + bit, s = g / 12, r = g - 12 * s;
      uint32_t bv = 0;
      if (s < n)
        bv = ((uint32_t)idx[s] >> (11 - r)) & 1u;
      else if (g == n * 12 + 1)
        bv = 1u;  // torchac terminator "01"
      v = (v << 1) | bv;
    }
    out[o] = (uint8_t)v;
  }
}

__global__ void unpack12_kernel(const uint8_t *__restrict__ in, int n, int nbytes, int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  in += (size_t)b * nbytes;
  idx += (size_t)b * n;
  for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
    const int g = s * 12, o = g >> 3;
    uint32_t w = ((uint32_t)in[o] << 16) | ((uint32_t)in[o + 1] << 8) | (o + 2 < nbytes ? (uint32_t)in[o + 2] : 0u);
    idx[s] = (int32_t)((w >> (24 - 12 - (g & 7))) & 0xfffu);
  }
}

extern "C" int sgic_pack12_batch(const int32_t *d_idx, int B, int n, uint8_t *d_out, sgic_stream_t stream) {
  SGIC_REQUIRE(d_idx && d_out && B > 0 && n >= 0, "args");
  const int nb = (int)sgic_pack12_size(n);
  pack12_kernel<<<dim3(cdiv(nb, 64), B), 64, 0, to_stream(stream)>>>(d_idx, n, d_out, nb);
  return sgic::check_launch("pack12_kernel");
}

extern "C" int sgic_unpack12_batch(const uint8_t *d_in, int B, int n, int32_t *d_idx, sgic_stream_t stream) {
  SGIC_REQUIRE(d_in && d_idx && B > 0 && n > 0, "args");
  const int nb = (int)sgic_pack12_size(n);
  unpack12_kernel<<<dim3(cdiv(n, 64), B), 64, 0, to_stream(stream)>>>(d_in, n, nb, d_idx);
  return sgic::check_launch("unpack12_kernel");
}

// ------------------------------------------------------------------------------------------------
// 4-step masked quantiser + index builder, NHWC.  One thread per (image, position, c16).
// Quarter q of the channels is active at spatial phase p = 2*(i&1)+(j&1) iff q == p ^ xk[k]
// (compression_model.py:277-280).  Compiled with -ffp-contract=off: every float op is a single IEEE
// operation so the results equal the CPU oracle bit for bit.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int active_quarter(int i, int j, int k) {
  const int p = ((i & 1) << 1) | (j & 1);
  const int xk = (4 - k) & 3;  // {0,3,2,1}
  return p ^ xk;
}

__device__ __forceinline__ int scale_to_index(float sg_hat, float thr) {
  const float log_min = (float)-2.2072749131897207;         // ln 0.11
  const float log_step = (float)0.024965325476664284;       // (ln 64 - ln 0.11) / 255
  const float sc = fmaxf(sg_hat, 1e-5f);
  float fi = ((float)log((double)sc) - log_min) / log_step;  // correctly rounded logf, then fp32 ops
  fi = fminf(fmaxf(fi, 0.f), 255.f);
  int ii = (int)fi;
  if (thr >= 0.f && sg_hat < thr) ii = -1;
  return ii;
}

template <int MODE>  // 0: quantise (encoder), 1: indexes only (decoder), 2: dequant (decoder)
__global__ void quant_step_kernel(const float *__restrict__ y, const float *__restrict__ scales,
                                  const float *__restrict__ means, int ld_sm, float *__restrict__ yhat, int ld_yhat,
                                  int B, int H, int W, int C, int k, float thr, int16_t *__restrict__ sym,
                                  int16_t *__restrict__ idx) {
  const int Q = C >> 2;
  const long total = (long)B * H * W * Q;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % Q);
    const long pos = t / Q;  // b*H*W + i*W + j
    const int j = (int)(pos % W);
    const int i = (int)((pos / W) % H);
    const int b = (int)(pos / ((long)W * H));
    const int ch = active_quarter(i, j, k) * Q + c;
    const size_t o = ((((size_t)b * 4 + k) * Q + c) * H + i) * W + j;
    if (MODE == 0) {
      const float mu = means[pos * ld_sm + ch], sg = scales[pos * ld_sm + ch];
      float s = rintf(y[pos * C + ch] - mu);  // round half to even
      float sg_hat = sg;
      if (thr >= 0.f && sg < thr) {
        s = 0.f;
        sg_hat = 0.f;
      }
      yhat[pos * ld_yhat + ch] = s + mu;
      sym[o] = (int16_t)fminf(fmaxf(s, -30000.f), 30000.f);
      idx[o] = (int16_t)scale_to_index(sg_hat, thr);
    } else if (MODE == 1) {
      const float sg = scales[pos * ld_sm + ch];
      const float sg_hat = (thr >= 0.f && sg < thr) ? 0.f : sg;
      idx[o] = (int16_t)scale_to_index(sg_hat, thr);
    } else {
      const float mu = means[pos * ld_sm + ch];
      yhat[pos * ld_yhat + ch] = (float)sym[o] + mu;
    }
  }
}

// flat GaussianEncoder.build_indexes (entropy_models.py:355-362): idx = trunc(clamp((ln max(s,1e-5) - ln .11)/step, 0, 255)),
// -1 where max(s,1e-5) < thr
__global__ void scale_to_index_kernel(const float *__restrict__ scales, long n, float thr, int16_t *__restrict__ idx) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    idx[i] = (int16_t)scale_to_index(scales[i], thr);
}

extern "C" int sgic_scale_to_index(const float *d_scales, long n, float thr, int16_t *d_idx, sgic_stream_t stream) {
  SGIC_REQUIRE(d_scales && d_idx && n > 0, "args");
  scale_to_index_kernel<<<cdiv(n, 256) > 4096 ? 4096 : cdiv(n, 256), 256, 0, to_stream(stream)>>>(d_scales, n, thr, d_idx);
  return sgic::check_launch("scale_to_index_kernel");
}

extern "C" int sgic_quant_step(const float *d_y, const float *d_scales, const float *d_means, int ld_sm,
                               float *d_yhat, int ld_yhat, int B, int H, int W, int C, int k, float thr,
                               int16_t *d_sym, int16_t *d_idx, sgic_stream_t stream) {
  SGIC_REQUIRE(d_y && d_scales && d_means && d_yhat && d_sym && d_idx, "null");
  SGIC_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k >= 0 && k < 4 && ld_sm >= C && ld_yhat >= C, "shape");
  const long total = (long)B * H * W * (C / 4);
  quant_step_kernel<0><<<cdiv(total, 256), 256, 0, to_stream(stream)>>>(d_y, d_scales, d_means, ld_sm, d_yhat,
                                                                         ld_yhat, B, H, W, C, k, thr, d_sym, d_idx);
  return sgic::check_launch("quant_step_kernel<0>");
}

extern "C" int sgic_index_step(const float *d_scales, int ld_sm, int B, int H, int W, int C, int k, float thr,
                               int16_t *d_idx, sgic_stream_t stream) {
  SGIC_REQUIRE(d_scales && d_idx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k >= 0 && k < 4 && ld_sm >= C,
               "args");
  const long total = (long)B * H * W * (C / 4);
  quant_step_kernel<1><<<cdiv(total, 256), 256, 0, to_stream(stream)>>>(nullptr, d_scales, nullptr, ld_sm, nullptr, 0,
                                                                         B, H, W, C, k, thr, nullptr, d_idx);
  return sgic::check_launch("quant_step_kernel<1>");
}

// Decision margins of the index builder for step k: how far (in units of one index step, i.e. of ln(sigma) / log_step) each
// coded position is from the nearest decision boundary of build_indexes -- a bin edge of the 256-level scale table or the
// force-zero / skip threshold -- and the index the OTHER side of that boundary would give.  A decoder that reads a
// stream made by a different fp32 implementation (the reference on a CPU / another GPU) sees sigma with a few ulps of
// summation-order noise; a sigma sitting within that noise of a boundary can flip an index and desynchronise rANS.  The
// host uses these margins to retry such a decode with the near-boundary index flipped (bottleneck.py), which the rANS
// end-of-stream condition then verifies.  d_margin / d_alt: (B, 4, C/4, H, W), step-k slice written.
__global__ void index_margin_kernel(const float *__restrict__ scales, int ld_sm, int B, int H, int W, int C, int k, float thr,
                                    float *__restrict__ margin, int16_t *__restrict__ alt) {
  const int Q = C >> 2;
  const long total = (long)B * H * W * Q;
  const double log_min = -2.2072749131897207, log_step = 0.024965325476664284;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c = (int)(t % Q);
    const long pos = t / Q;
    const int j = (int)(pos % W);
    const int i = (int)((pos / W) % H);
    const int b = (int)(pos / ((long)W * H));
    const int ch = active_quarter(i, j, k) * Q + c;
    const size_t o = ((((size_t)b * 4 + k) * Q + c) * H + i) * W + j;
    const float sg = scales[pos * ld_sm + ch];
    const bool skipped = thr >= 0.f && sg < thr;
    const double ls = log((double)fmaxf(sg, 1e-5f));
    const int as_coded = scale_to_index(sg, -1.f);          // the index this sigma gets when it is not skipped
    double best = 1e30;
    int other = as_coded;
    if (thr >= 0.f) {                                        // the skip threshold
      best = fabs(ls - log((double)thr)) / log_step;
      other = skipped ? as_coded : -1;
    }
    if (!skipped) {                                          // bin edges (none below index 0 / above 255: the clamp)
      const double ti = (ls - log_min) / log_step, f = ti - floor(ti);
      if (ti > 0.0 && ti < 255.0) {
        if (as_coded >= 1 && f < best) best = f, other = as_coded - 1;
        if (as_coded <= 254 && 1.0 - f < best) best = 1.0 - f, other = as_coded + 1;
      }
    }
    margin[o] = (float)best;
    alt[o] = (int16_t)other;
  }
}

extern "C" int sgic_index_margins(const float *d_scales, int ld_sm, int B, int H, int W, int C, int k, float thr,
                                  float *d_margin, int16_t *d_alt, sgic_stream_t stream) {
  SGIC_REQUIRE(d_scales && d_margin && d_alt && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k >= 0 && k < 4 && ld_sm >= C,
               "args");
  const long total = (long)B * H * W * (C / 4);
  index_margin_kernel<<<cdiv(total, 256), 256, 0, to_stream(stream)>>>(d_scales, ld_sm, B, H, W, C, k, thr, d_margin, d_alt);
  return sgic::check_launch("index_margin_kernel");
}

extern "C" int sgic_dequant_step(const int16_t *d_sym, const float *d_means, int ld_sm, float *d_yhat, int ld_yhat,
                                 int B, int H, int W, int C, int k, sgic_stream_t stream) {
  SGIC_REQUIRE(d_sym && d_means && d_yhat && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && k >= 0 && k < 4 &&
                   ld_sm >= C && ld_yhat >= C,
               "args");
  const long total = (long)B * H * W * (C / 4);
  quant_step_kernel<2><<<cdiv(total, 256), 256, 0, to_stream(stream)>>>(nullptr, nullptr, d_means, ld_sm, d_yhat,
                                                                         ld_yhat, B, H, W, C, k, -1.f,
                                                                         const_cast<int16_t *>(d_sym), nullptr);
  return sgic::check_launch("quant_step_kernel<2>");
}
