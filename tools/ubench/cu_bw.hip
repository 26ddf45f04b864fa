// Per-CU read throughput for the access shapes of the recurrent-step kernels (data L2/MALL resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: fragment-shaped: lane (r=lane&31,h=lane>>5) reads 16 B at row (rowbase + r), k offset (16*j + 8*h) bf16  -> 32 rows x 32 B per instruction
// mode 1: full-line: lane reads 16 B at row (rowbase + lane>>3), offset (lane&7)*16 B + 128*j                      -> 8 rows x 128 B per instruction
template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ w, int rows_per_block, int rowbytes, uint4* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = (const char*)w + (size_t)blockIdx.x * rows_per_block * rowbytes;
  uint4 acc = make_uint4(0, 0, 0, 0);
  // each wave streams a K quarter of every row of the block: rows_per_block rows x rowbytes/4 bytes
  const int kq = rowbytes / 4, k0 = wave * kq;
  if (MODE == 0) {
    for (int rb = 0; rb < rows_per_block; rb += 32) {
      const char* p = base + (size_t)(rb + (lane & 31)) * rowbytes + k0 + (lane >> 5) * 16;
      for (int j = 0; j < kq; j += 32 * DEPTH) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const uint4*)(p + j + 32 * d);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
      }
    }
  } else {
    for (int rb = 0; rb < rows_per_block; rb += 8 * DEPTH) {
      for (int j = 0; j < kq; j += 128) {
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = *(const uint4*)(base + (size_t)(rb + 8 * d + (lane >> 3)) * rowbytes + k0 + j + (lane & 7) * 16);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y ^= v[d].y; acc.z ^= v[d].z; acc.w ^= v[d].w; }
      }
    }
  }
  if (acc.x == 0x12345678u) out[threadIdx.x] = acc;
}

template <int MODE, int DEPTH> static int run(const uint4* w, uint4* out, int blocks, int rows, int rowbytes, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((rd<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, w, rows, rowbytes, out);
  CK(hipEventRecord(e0));
  const int it = 50;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((rd<MODE, DEPTH>), dim3(blocks), dim3(256), 0, 0, w, rows, rowbytes, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / it, bytes = (double)blocks * rows * rowbytes;
  printf("%-28s blocks %4d  %6.0f KB/block  %7.2f us  %7.1f GB/s per block  %6.2f TB/s total\n", name, blocks, rows * rowbytes / 1024.0, us,
         rows * (double)rowbytes / us / 1e3, bytes / us / 1e6);
  return 0;
}

int main() {
  const int rowbytes = 2048;                       // K = 1024 bf16
  size_t total = (size_t)512 * 160 * rowbytes;     // up to 512 blocks x 160 rows
  uint4 *w, *out; CK(hipMalloc(&w, total)); CK(hipMalloc(&out, 1 << 20)); CK(hipMemset(w, 1, total));
  for (int blocks : {64, 128, 256, 512}) {
    run<0, 1>(w, out, blocks, 160, rowbytes, "fragment depth1");
    run<0, 4>(w, out, blocks, 160, rowbytes, "fragment depth4");
    run<0, 8>(w, out, blocks, 160, rowbytes, "fragment depth8");
    run<1, 1>(w, out, blocks, 160, rowbytes, "full-line depth1");
    run<1, 4>(w, out, blocks, 160, rowbytes, "full-line depth4");
    run<1, 10>(w, out, blocks, 160, rowbytes, "full-line depth10");
  }
  return 0;
}
