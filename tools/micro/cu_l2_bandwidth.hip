// Diagnostic: what can ONE CU pull from L2, and does the access shape matter?  (round 3: the small-tile GEMMs of a single image and
// the 128x256 tile both sit at ~40-55 GB/s per CU of operand traffic.)  Each workgroup re-reads its own 64 KiB region (L2 resident:
// 32 workgroups x 64 KiB per XCD) with 16-byte loads per lane in one of four shapes per wave instruction:
//   0: 1 KiB contiguous            1: 16 rows x 64 B  (row stride 2 KiB: the GEMM's piece, half a cache line per row)
//   2: 8 rows x 128 B (full lines) 3: 4 rows x 256 B
// as plain global_load_dwordx4 (registers) or as global_load_lds_dwordx4 (LDS-DMA), with 4 or 8 waves per workgroup and 1 or 2
// workgroups per CU.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o tools/micro/bin/cu_l2_bandwidth tools/micro/cu_l2_bandwidth.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

template <int SHAPE, bool DMA>
__global__ __launch_bounds__(512) void pull(const char *buf, unsigned *sink, int iters, int region) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const char *base = buf + (size_t)blockIdx.x * region;
  // a wave instruction covers 1 KiB of the region "tile" of 16 rows x 2 KiB (32 KiB); instruction j of the tile:
  unsigned off;
  if (SHAPE == 0) off = lane * 16;                                   // + j * 1024
  if (SHAPE == 1) off = (lane >> 2) * 2048 + (lane & 3) * 16;        // + j * 64      (j < 32)
  if (SHAPE == 2) off = (lane >> 3) * 2048 + (lane & 7) * 16;        // rows 8 (j & 1) .., + (j >> 1) * 128
  if (SHAPE == 3) off = (lane >> 4) * 2048 + (lane & 15) * 16;       // rows 4 (j & 3) .., + (j >> 2) * 256
  u32x4 acc = {0, 0, 0, 0};
  const int tiles = region / 32768;
  for (int it = 0; it < iters; it++) {
    for (int t = 0; t < tiles; t++) {
      const char *tb = base + t * 32768;
#pragma unroll 8
      for (int j = wave; j < 32; j += nw) {
        unsigned o;
        if (SHAPE == 0) o = off + j * 1024;
        if (SHAPE == 1) o = off + j * 64;
        if (SHAPE == 2) o = off + (j & 1) * 8 * 2048 + (j >> 1) * 128;
        if (SHAPE == 3) o = off + (j & 3) * 4 * 2048 + (j >> 2) * 256;
        if (DMA) {
          __builtin_amdgcn_global_load_lds((glb_void *)(tb + o), (lds_void *)(smem + ((j * 1024) & 32767)), 16, 0, 0);
        } else {
          const u32x4 v = *reinterpret_cast<const u32x4 *>(tb + o);
          acc ^= v;
        }
      }
    }
  }
  if (DMA) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<u32x4 *>(smem + threadIdx.x * 16);
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int SHAPE, bool DMA>
static void run(const char *buf, unsigned *sink, int wgs, int threads, int region, const char *label) {
  const int iters = 200;
  hipFuncSetAttribute(reinterpret_cast<const void *>(pull<SHAPE, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int lds = DMA ? 32768 : 0;
  pull<SHAPE, DMA><<<wgs, threads, lds>>>(buf, sink, 20, region);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  pull<SHAPE, DMA><<<wgs, threads, lds>>>(buf, sink, iters, region);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * region * iters;
  printf("  %-34s %s: %7.1f GB/s per workgroup, %6.2f TB/s chip-wide\n", label, DMA ? "LDS-DMA  " : "registers", bytes / wgs / ms / 1e6, bytes / ms / 1e9);
}

int main() {
  const int region = 65536;
  char *buf;
  unsigned *sink;
  hipMalloc(&buf, (size_t)512 * region);
  hipMalloc(&sink, 4);
  hipMemset(buf, 1, (size_t)512 * region);
  for (int wgs : {256, 512, 64}) {
    for (int threads : {256, 512}) {
      printf("%d workgroups x %d waves, each re-reading its own %d KiB:\n", wgs, threads / 64, region / 1024);
      run<0, false>(buf, sink, wgs, threads, region, "1 KiB contiguous");
      run<1, false>(buf, sink, wgs, threads, region, "16 rows x 64 B (stride 2 KiB)");
      run<2, false>(buf, sink, wgs, threads, region, "8 rows x 128 B");
      run<3, false>(buf, sink, wgs, threads, region, "4 rows x 256 B");
      run<0, true>(buf, sink, wgs, threads, region, "1 KiB contiguous");
      run<1, true>(buf, sink, wgs, threads, region, "16 rows x 64 B (stride 2 KiB)");
      run<2, true>(buf, sink, wgs, threads, region, "8 rows x 128 B");
      run<3, true>(buf, sink, wgs, threads, region, "4 rows x 256 B");
    }
  }
  return 0;
}
