// Baseline JPEG decode on the GPU for the compress driver's ingest (replaces the pixel decode inside the reference's
// Test_Dataset, compress.py:151-168: `Image.open(path).convert("RGB")`), BIT-EXACT with Pillow / libjpeg-turbo's default decoder:
// Huffman decode -> dequantise + "islow" integer IDCT (jidctint.c) -> fancy (triangle) chroma upsampling (jdsample.c: h2v1, h2v2,
// h1v2) -> YCbCr -> RGB with the fixed-point tables of jdcolor.c -> u8 HWC, the layout sgic_u8hwc_to_f32chw_pad consumes.
// The host (sgic_amd/jpeg.py) only parses markers, builds the Huffman lookup tables and strips the 0xFF00 stuffing / RSTn markers
// (byte shuffling, no entropy decoding); every coded bit is decoded here.
//
// Three kernels over a batch:
//   jpeg_huff_kernel      one wave per image.  The entropy-coded segment is a serial bit stream, so the wave runs the decode loop as
//                         wave-uniform code (every lane computes the same state -- the cost of one lane) and uses its 64 lanes for
//                         what IS parallel: streaming the scan through a 4 KiB LDS ring in 1 KiB coalesced chunks, zeroing /
//                         storing each 64-coefficient block (lane = coefficient), holding the lookup tables in LDS.
//   jpeg_idct_kernel      one thread per 8x8 block: dequantise, two 1-D passes of the LL&M integer IDCT, range limit -> u8 planes.
//   jpeg_color_kernel     one thread per output pixel: chroma upsampling of the 4:4:4 / 4:2:2 / 4:2:0 / 4:4:0 layouts + colour
//                         conversion (or grey -> RGB).
// All integer arithmetic: parity with Pillow is exact (tests/test_gpu_jpeg.py), not a tolerance.
#include "common.h"

#define JPG_NP 64            // int32 parameters per image (sgic_amd/jpeg.py: PARAM_*)
#define JPG_LOOK 9           // bits of the fast Huffman lookup
#define JPG_TAB_BYTES 1424   // one table: u16 fast[512] | i32 maxcode[18] | i32 valoff[17] | pad 4 | u8 huffval[256]
#define JPG_RING 4096     // LDS: ring 4 KiB + tables 5.6 KiB = 9.9 KiB: the wave fits BESIDE a 144 KiB GEMM workgroup on the same CU
#define JPG_CHUNK 1024

enum {
  P_SCAN_OFF = 0, P_SCAN_LEN, P_TAB_OFF, P_QUANT_OFF, P_NCOMP, P_W, P_H, P_HMAX, P_VMAX, P_MCUS_X, P_MCUS_Y, P_RESTART, P_SEG_OFF, P_NSEG,
  P_COMP0 = 16,   // per component, 12 ints: h, v, qtab, dctab, actab, blocks_w, blocks_h, comp_w, comp_h, coef_off (blocks), plane_off (bytes), unused
  P_CSTRIDE = 12
};

__constant__ unsigned char jpg_natural[80] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                              6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                              39, 46, 53, 60, 61, 54, 47, 55, 62, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

// ---- Huffman decode: one wave per image -------------------------------------------------------------------------------------------
// params / scan / tabs / segs / quant_in may live in PINNED HOST memory (the kernel pulls them over PCIe: a batch is ~1 MB, read once,
// in coalesced 1 KiB chunks) -- there is then no H2D copy to schedule; the descriptors and quantisation tables the two later kernels
// need are left in device memory (params_dev, quant_dev) by this one.
__global__ __launch_bounds__(64) void jpeg_huff_kernel(const int *__restrict__ params, const unsigned char *__restrict__ scan,
                                                       const unsigned char *__restrict__ tabs, const int *__restrict__ segs,
                                                       const unsigned short *__restrict__ quant_in, int *__restrict__ params_dev,
                                                       unsigned short *__restrict__ quant_dev, short *__restrict__ coef,
                                                       int *__restrict__ err) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[JPG_RING];
  __shared__ __attribute__((aligned(16))) unsigned char tab[4 * JPG_TAB_BYTES];
  __shared__ short blk[64];
  const int img = blockIdx.x, lane = threadIdx.x;
  __shared__ int Ps[JPG_NP];
  Ps[lane] = params[(size_t)img * JPG_NP + lane];
  params_dev[(size_t)img * JPG_NP + lane] = Ps[lane];
  __syncthreads();
  const int *P = Ps;
  for (int i = lane; i < 256; i += 64) quant_dev[(size_t)img * 256 + i] = quant_in[P[P_QUANT_OFF] + i];
  const unsigned char *src = scan + P[P_SCAN_OFF];
  const int total = P[P_SCAN_LEN];   // bytes, padded by the host to a multiple of JPG_CHUNK with zeros
  {
    const uint4 *t4 = reinterpret_cast<const uint4 *>(tabs + P[P_TAB_OFF]);
    for (int i = lane; i < 4 * JPG_TAB_BYTES / 16; i += 64) reinterpret_cast<uint4 *>(tab)[i] = t4[i];
  }
  int filled = 0;   // bytes of the scan staged so far (wave-uniform)
  auto fill = [&]() {
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + filled);
    uint4 *d4 = reinterpret_cast<uint4 *>(ring + (filled & (JPG_RING - 1)));
#pragma unroll
    for (int i = 0; i < JPG_CHUNK / 16 / 64; i++) d4[lane + 64 * i] = s4[lane + 64 * i];
    filled += JPG_CHUNK;
  };
  fill();
  if (total > JPG_CHUNK) fill();
  __syncthreads();
  // bit reader: `acc` holds `cnt` valid bits, MSB first; rp = byte position of the next 32-bit word to fetch
  unsigned long long acc = 0;
  int cnt = 0, rp = 0, bad = 0;
  auto refill = [&]() {
    while (cnt <= 32) {
      if (filled - rp < JPG_CHUNK + 8 && filled < total) {   // uniform: stage the next chunk (the ring keeps > 4 KiB unread at most)
        fill();
        __syncthreads();
      }
      unsigned w = 0;
      if (rp < total) w = *reinterpret_cast<const unsigned *>(ring + (rp & (JPG_RING - 1)));
      w = __builtin_bswap32(w);
      acc |= (unsigned long long)w << (32 - cnt);
      cnt += 32;
      rp += 4;
    }
  };
  auto seek = [&](int byte_off) {   // restart boundary: continue at a byte offset of the cleaned scan
    const int aligned = byte_off & ~3;
    if (aligned >= filled || aligned < filled - JPG_RING + JPG_CHUNK) {   // outside the staged window: restage from there
      filled = aligned & ~(JPG_CHUNK - 1);
      fill();
      if (filled < total) fill();
      __syncthreads();
    }
    rp = aligned;
    acc = 0;
    cnt = 0;
    refill();
    const int skip = (byte_off - aligned) * 8;
    acc <<= skip;
    cnt -= skip;
  };
  auto decode = [&](const unsigned char *T) -> int {   // one Huffman symbol
    refill();
    const unsigned e = reinterpret_cast<const unsigned short *>(T)[(unsigned)(acc >> (64 - JPG_LOOK))];
    if (e) {
      const int nb = e >> 8;
      acc <<= nb;
      cnt -= nb;
      return e & 255;
    }
    const int *maxcode = reinterpret_cast<const int *>(T + 1024), *valoff = maxcode + 18;
    const unsigned code16 = (unsigned)(acc >> 48);
    int l = JPG_LOOK + 1;
    while (l <= 16 && (int)(code16 >> (16 - l)) > maxcode[l]) l++;
    if (l > 16) {
      bad = 1;
      return 0;
    }
    const int sym = T[1024 + 72 + 68 + 4 + ((valoff[l] + (int)(code16 >> (16 - l))) & 255)];
    acc <<= l;
    cnt -= l;
    return sym;
  };
  auto receive = [&](int s) -> int {   // s extra bits, sign-extended (jdhuff.c HUFF_EXTEND)
    const int v = (int)(acc >> (64 - s));
    acc <<= s;
    cnt -= s;
    return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
  };

  const int ncomp = P[P_NCOMP], mcus = P[P_MCUS_X] * P[P_MCUS_Y], mcus_x = P[P_MCUS_X], restart = P[P_RESTART];
  const int *seg = segs + P[P_SEG_OFF];
  int pred[3] = {0, 0, 0};
  int next_seg = 1;
  for (int mcu = 0; mcu < mcus && !bad; mcu++) {
    if (restart && mcu && mcu % restart == 0) {
      if (next_seg >= P[P_NSEG]) {
        bad = 2;
        break;
      }
      seek(seg[next_seg++]);
      pred[0] = pred[1] = pred[2] = 0;
    }
    const int my = mcu / mcus_x, mx = mcu - my * mcus_x;
    for (int c = 0; c < ncomp; c++) {
      const int *C = P + P_COMP0 + c * P_CSTRIDE;
      const int hs = ncomp == 1 ? 1 : C[0], vs = ncomp == 1 ? 1 : C[1];   // a single-component scan is not interleaved: one block per MCU
      const unsigned char *Tdc = tab + C[3] * JPG_TAB_BYTES, *Tac = tab + (2 + C[4]) * JPG_TAB_BYTES;
      for (int by = 0; by < vs; by++)
        for (int bx = 0; bx < hs; bx++) {
          blk[lane] = 0;
          int s = decode(Tdc);
          if (s) {
            refill();
            pred[c] += receive(s);
          }
          if (lane == 0) blk[0] = (short)pred[c];
          for (int k = 1; k < 64;) {
            const int rs = decode(Tac);
            const int r = rs >> 4;
            s = rs & 15;
            if (s == 0) {
              if (r != 15) break;   // EOB
              k += 16;
              continue;
            }
            k += r;
            refill();
            const int v = receive(s);
            if (lane == 0) blk[jpg_natural[k]] = (short)v;   // k <= 63 + 15: the table is padded to 80 entries
            k++;
          }
          const size_t b = (size_t)C[9] + (size_t)(my * vs + by) * C[5] + (mx * hs + bx);
          coef[b * 64 + lane] = blk[lane];
        }
    }
  }
  if (lane == 0) err[img] = bad;
}

// ---- dequantise + islow IDCT (jidctint.c jpeg_idct_islow), one thread per block ------------------------------------------------------
#define JC(x) ((int)(x))
#define J_DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
__device__ __forceinline__ void jpg_idct_1d(const int in[8], int out[8], int shift) {
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * JC(4433);
  int tmp2 = z1 + z3 * -JC(15137);
  int tmp3 = z1 + z2 * JC(6270);
  z2 = in[0];
  z3 = in[4];
  int tmp0 = (z2 + z3) << 13;
  int tmp1 = (z2 - z3) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7];
  tmp1 = in[5];
  tmp2 = in[3];
  tmp3 = in[1];
  z1 = tmp0 + tmp3;
  z2 = tmp1 + tmp2;
  z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = (z3 + z4) * JC(9633);
  tmp0 *= JC(2446);
  tmp1 *= JC(16819);
  tmp2 *= JC(25172);
  tmp3 *= JC(12299);
  z1 *= -JC(7373);
  z2 *= -JC(20995);
  z3 *= -JC(16069);
  z4 *= -JC(3196);
  z3 += z5;
  z4 += z5;
  tmp0 += z1 + z3;
  tmp1 += z2 + z4;
  tmp2 += z2 + z3;
  tmp3 += z1 + z4;
  out[0] = J_DESCALE(tmp10 + tmp3, shift);
  out[7] = J_DESCALE(tmp10 - tmp3, shift);
  out[1] = J_DESCALE(tmp11 + tmp2, shift);
  out[6] = J_DESCALE(tmp11 - tmp2, shift);
  out[2] = J_DESCALE(tmp12 + tmp1, shift);
  out[5] = J_DESCALE(tmp12 - tmp1, shift);
  out[3] = J_DESCALE(tmp13 + tmp0, shift);
  out[4] = J_DESCALE(tmp13 - tmp0, shift);
}

__device__ __forceinline__ unsigned char jpg_range_limit(int v) {   // libjpeg's range_limit table behind `& RANGE_MASK`, centred at 128
  const int i = v & 1023;
  return (unsigned char)(i < 128 ? i + 128 : (i < 512 ? 255 : (i < 896 ? 0 : i - 896)));
}

__global__ void jpeg_idct_kernel(const int *__restrict__ params, const short *__restrict__ coef, const unsigned short *__restrict__ quant,
                                 unsigned char *__restrict__ planes, int max_blocks) {
  const int img = blockIdx.y;
  const int *P = params + (size_t)img * JPG_NP;
  const int ncomp = P[P_NCOMP];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < max_blocks; t += gridDim.x * blockDim.x) {
    int c = 0, local = t;
    while (c < ncomp) {   // which component does block t of this image belong to?
      const int nb = P[P_COMP0 + c * P_CSTRIDE + 5] * P[P_COMP0 + c * P_CSTRIDE + 6];
      if (local < nb) break;
      local -= nb;
      c++;
    }
    if (c >= ncomp) continue;
    const int *C = P + P_COMP0 + c * P_CSTRIDE;
    const int bw = C[5], by = local / bw, bx = local - by * bw;
    const short *cf = coef + ((size_t)C[9] + local) * 64;
    const unsigned short *q = quant + (size_t)img * 256 + C[2] * 64;   // quant_dev of jpeg_huff_kernel: image-major
    int ws[64];
    // pass 1: columns
#pragma unroll
    for (int x = 0; x < 8; x++) {
      int in[8], o[8];
#pragma unroll
      for (int y = 0; y < 8; y++) in[y] = (int)cf[y * 8 + x] * (int)q[y * 8 + x];
      jpg_idct_1d(in, o, 13 - 2);
#pragma unroll
      for (int y = 0; y < 8; y++) ws[y * 8 + x] = o[y];
    }
    // pass 2: rows
    unsigned char *dst = planes + (size_t)C[10] + (size_t)(by * 8) * (bw * 8) + bx * 8;
#pragma unroll
    for (int y = 0; y < 8; y++) {
      int o[8];
      jpg_idct_1d(ws + y * 8, o, 13 + 2 + 3);
      unsigned lo = 0, hi = 0;
#pragma unroll
      for (int x = 0; x < 4; x++) {
        lo |= (unsigned)jpg_range_limit(o[x]) << (8 * x);
        hi |= (unsigned)jpg_range_limit(o[x + 4]) << (8 * x);
      }
      *reinterpret_cast<uint2 *>(dst + (size_t)y * (bw * 8)) = make_uint2(lo, hi);
    }
  }
}

// ---- chroma upsampling (jdsample.c fancy upsamplers) + YCbCr -> RGB (jdcolor.c) -----------------------------------------------------
__device__ __forceinline__ int jpg_chroma(const unsigned char *pl, int stride, int cw, int ch, int hs, int vs, int x, int y) {
  // value of a chroma plane (downsampled cw x ch, row stride `stride`) at full-resolution pixel (x, y); hs / vs = luma samples per chroma
  // sample horizontally / vertically (1 or 2)
  if (hs == 1 && vs == 1) return pl[(size_t)y * stride + x];
  if (vs == 1) {   // h2v1_fancy_upsample
    const unsigned char *r = pl + (size_t)y * stride;
    const int c = x >> 1;
    if (x & 1) return c == cw - 1 ? r[c] : (3 * r[c] + r[c + 1] + 2) >> 2;
    return c == 0 ? r[0] : (3 * r[c] + r[c - 1] + 1) >> 2;
  }
  const int cy = y >> 1, v = y & 1;
  const int fy = v ? min(cy + 1, ch - 1) : max(cy - 1, 0);
  const unsigned char *r0 = pl + (size_t)cy * stride, *r1 = pl + (size_t)fy * stride;
  if (hs == 1) return (3 * r0[x] + r1[x] + (v ? 2 : 1)) >> 2;   // h1v2_fancy_upsample
  // h2v2_fancy_upsample: 9/16, 3/16, 3/16, 1/16
  const int c = x >> 1;
  const int cur = 3 * r0[c] + r1[c];
  if (x & 1) {
    if (c == cw - 1) return (cur * 4 + 7) >> 4;
    return (cur * 3 + 3 * r0[c + 1] + r1[c + 1] + 7) >> 4;
  }
  if (c == 0) return (cur * 4 + 8) >> 4;
  return (cur * 3 + 3 * r0[c - 1] + r1[c - 1] + 8) >> 4;
}

__global__ void jpeg_color_kernel(const int *__restrict__ params, const unsigned char *__restrict__ planes, unsigned char *__restrict__ out,
                                  int H, int W) {
  const int img = blockIdx.y;
  const int *P = params + (size_t)img * JPG_NP;
  const int ncomp = P[P_NCOMP];
  const int *C0 = P + P_COMP0, *C1 = C0 + P_CSTRIDE, *C2 = C1 + P_CSTRIDE;
  const int hmax = P[P_HMAX], vmax = P[P_VMAX];
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)H * W; t += (long)gridDim.x * blockDim.x) {
    const int y = (int)(t / W), x = (int)(t - (long)y * W);
    const int yy = planes[(size_t)C0[10] + (size_t)y * (C0[5] * 8) + x];
    int r = yy, g = yy, b = yy;
    if (ncomp == 3) {
      const int hs = hmax / C1[0], vs = vmax / C1[1];
      const int cb = jpg_chroma(planes + C1[10], C1[5] * 8, C1[7], C1[8], hs, vs, x, y) - 128;
      const int cr = jpg_chroma(planes + C2[10], C2[5] * 8, C2[7], C2[8], hs, vs, x, y) - 128;
      // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, ONE_HALF 32768, FIX(1.40200) 91881, FIX(1.77200) 116130, FIX(0.71414) 46802, FIX(0.34414) 22554
      r = yy + ((91881 * cr + 32768) >> 16);
      b = yy + ((116130 * cb + 32768) >> 16);
      g = yy + ((-22554 * cb + 32768 + -46802 * cr) >> 16);
      r = min(max(r, 0), 255);
      g = min(max(g, 0), 255);
      b = min(max(b, 0), 255);
    }
    unsigned char *o = out + ((size_t)img * H * W + (size_t)t) * 3;
    o[0] = (unsigned char)r;
    o[1] = (unsigned char)g;
    o[2] = (unsigned char)b;
  }
}

// Decode a batch of B baseline JPEGs of equal geometry (H, W) to RGB u8 HWC (B, H, W, 3) on the device.
//   d_params  B x 64 int32 (layout above, built by sgic_amd/jpeg.py), d_scan the cleaned entropy-coded segments (each padded to a
//   multiple of 2048 bytes), d_tabs B x 4 Huffman tables (dc0, dc1, ac0, ac1) of 1424 bytes, d_segs the restart-interval byte offsets,
//   d_quant u16 quantisation tables in natural order, d_coef / d_planes workspaces (total_blocks * 64 int16 / plane_bytes u8),
//   d_err B int32 (0 ok, 1 invalid Huffman code, 2 missing restart segment).  max_blocks = the largest per-image block count.
//   The five input arrays may be device memory or device-accessible PINNED host memory (no copy is then needed at all);
//   d_work_params (B x 64 int32) / d_work_quant (B x 256 u16): device scratch for the copies the later kernels read.
extern "C" int sgic_jpeg_decode_batch(const int32_t *d_params, const uint8_t *d_scan, const uint8_t *d_tabs, const int32_t *d_segs,
                                      const uint16_t *d_quant, int32_t *d_work_params, uint16_t *d_work_quant, int16_t *d_coef,
                                      uint8_t *d_planes, uint8_t *d_out, int32_t *d_err, int B, int H, int W, int max_blocks,
                                      sgic_stream_t stream) {
  SGIC_REQUIRE(d_params && d_scan && d_tabs && d_segs && d_quant && d_work_params && d_work_quant && d_coef && d_planes && d_out && d_err, "null");
  SGIC_REQUIRE(B > 0 && H > 0 && W > 0 && max_blocks > 0, "shape");
  SGIC_REQUIRE((((uintptr_t)d_scan | (uintptr_t)d_tabs | (uintptr_t)d_planes) & 15) == 0, "16-byte alignment");
  hipStream_t st = to_stream(stream);
  jpeg_huff_kernel<<<B, 64, 0, st>>>(d_params, d_scan, d_tabs, d_segs, d_quant, d_work_params, d_work_quant, d_coef, d_err);
  int rc = sgic::check_launch("jpeg_huff_kernel");
  if (rc) return rc;
  jpeg_idct_kernel<<<dim3(cdiv(max_blocks, 64), B), 64, 0, st>>>(d_work_params, d_coef, d_work_quant, d_planes, max_blocks);
  rc = sgic::check_launch("jpeg_idct_kernel");
  if (rc) return rc;
  const long px = (long)H * W;
  jpeg_color_kernel<<<dim3((unsigned)min((px + 255) / 256, 4096L), B), 256, 0, st>>>(d_work_params, d_planes, d_out, H, W);
  return sgic::check_launch("jpeg_color_kernel");
}
