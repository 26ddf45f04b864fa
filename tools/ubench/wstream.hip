// Per-step cost of re-streaming a 512 KB bf16 weight matrix ([1024 gate columns][256 k]) from L2 into ONE workgroup, the
// access pattern of a persistent recurrent kernel that owns 16 batch rows for all T steps.
//  mode 0: MFMA 16x16x32 B-fragment shaped loads straight to VGPRs (16 rows x 64 B per wave-instruction)
//  mode 1: full 128-B lines (8 rows x 128 B per wave-instruction)
//  mode 2: LDS-DMA (global_load_lds_dwordx4, 8 rows x 128 B per wave-instruction) into a ring in LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void stream(const unsigned char* __restrict__ w, int steps, unsigned* out) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* base = w + (size_t)(blockIdx.x & 1) * 524288;             // two matrices (two directions)
  u32x4 acc = {0, 0, 0, 0};
  for (int st = 0; st < steps; ++st) {
    if (MODE == 0) {
      for (int i0 = 0; i0 < 64; i0 += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int i = i0 + d, n = i & 7, s = i >> 3;                             // tile n = (gate, sub), k-step s
          const int row = (n >> 1) * 256 + wave * 32 + (n & 1) * 16 + (lane & 15);
          v[d] = *(const u32x4*)(base + (size_t)row * 512 + 64 * s + 16 * (lane >> 4));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
      }
    } else if (MODE == 1) {
      for (int i0 = 0; i0 < 64; i0 += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
          const int i = i0 + d;                                                     // 64 instr x 1 KiB = this wave's 64 KB: rows wave*128 .. +128, 4 lines each
          const int row = wave * 128 + (i >> 2) * 8 + (lane >> 3);
          v[d] = *(const u32x4*)(base + (size_t)row * 512 + 128 * (i & 3) + 16 * (lane & 7));
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
      }
    } else {
      unsigned char* lb = lds + __builtin_amdgcn_readfirstlane(wave) * 8192;       // 8 KB per wave = 8 pieces in flight
      for (int i0 = 0; i0 < 64; i0 += 8) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const int i = i0 + d;
          const int row = wave * 128 + (i >> 2) * 8 + (lane >> 3);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)row * 512 + 128 * (i & 3) + 16 * (lane & 7)),
                                           (__attribute__((address_space(3))) void*)(lb + d * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x ^= *(const unsigned*)(lb + lane * 4);
      }
    }
    __builtin_amdgcn_s_barrier();
  }
  if (acc.x == 0x12345678u) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, int DEPTH> static int run(const unsigned char* w, unsigned* out, int blocks, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int steps = 63;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<MODE, DEPTH>), dim3(blocks), dim3(512), 0, 0, w, steps, out);
  CK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((stream<MODE, DEPTH>), dim3(blocks), dim3(512), 0, 0, w, steps, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us_step = ms * 1e3 / it / steps;
  printf("%-36s blocks %3d  %6.2f us per step  (%6.1f GB/s per workgroup)\n", name, blocks, us_step, 524288.0 / us_step / 1e3);
  return 0;
}

int main() {
  unsigned char* w; unsigned* out; CK(hipMalloc(&w, 2 * 524288)); CK(hipMalloc(&out, 4096)); CK(hipMemset(w, 1, 2 * 524288));
  for (int blocks : {2, 32, 64}) {
    run<0, 8>(w, out, blocks, "fragment 16 rows x 64 B, depth 8");
    run<0, 16>(w, out, blocks, "fragment 16 rows x 64 B, depth 16");
    run<0, 32>(w, out, blocks, "fragment 16 rows x 64 B, depth 32");
    run<1, 16>(w, out, blocks, "full lines, depth 16");
    run<1, 32>(w, out, blocks, "full lines, depth 32");
    run<2, 8>(w, out, blocks, "LDS-DMA full lines, 8 in flight/wave");
  }
  return 0;
}
