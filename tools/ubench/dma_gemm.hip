// Timing-only ablations of gemm_dma_bf16_kernel at the conv6-forward shape of workload C3
// (B=256, 4 x 64 map, 512 -> 512, 3x3): which of {LDS-DMA stream, fragment reads, MFMA issue} bounds the K loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch-attention-ocr_amd/csrc dma_gemm.hip -o dma_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mfma_gemm.h"
#include "epilogues.h"
using namespace aocr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void fill(bf16_t* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)(((h & 0xffff) / 32768.0f) - 1.0f);
  }
}

template <int ABL, bool PIPE = true, bool PAIRS = false> static int run(const LoadConvKh& a, const LoadKh& b, const EpConv& ep, int M, int N, int K, const bf16_t* zero, const char* name) {
  const int gx = N / 256, gy = (M + 255) / 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_dma_bf16_kernel<LoadConvKh, LoadKh, EpConv, ABL, PIPE, PAIRS>), dim3(gx * gy), dim3(512), 0, 0, a, b, ep, K, gx, gy, zero);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gemm_dma_bf16_kernel<LoadConvKh, LoadKh, EpConv, ABL, PIPE, PAIRS>), dim3(gx * gy), dim3(512), 0, 0, a, b, ep, K, gx, gy, zero);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / it;
  printf("%-44s %8.1f us  (%7.1f TFLOP/s if it were the full kernel)\n", name, us, 2.0 * M * N * K / us / 1e6);
  return 0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int B = 256, H = 4, W = 64, Cin = 512, Cout = 512, ks = 3, pad = 1, pool = 2;
  LoadConvK g; g.src = nullptr; g.H = H; g.W = W; g.C = Cin; g.KW = ks; g.sgn = 1; g.off = -pad; g.Hr = H; g.Wr = W;
  g.pmode = pool; g.Hp = H / 2; g.Wp = W / 2; g.rows = B * g.Hp * W * 2; g.K = ks * ks * Cin;
  const int M = g.rows, N = Cout, K = g.K;
  bf16_t *x, *w, *yb, *zero; float *y, *bias; uint8_t* idx;
  CK(hipMalloc(&x, (size_t)B * H * W * Cin * 2)); CK(hipMalloc(&w, (size_t)Cout * K * 2)); CK(hipMalloc(&zero, 64)); CK(hipMemset(zero, 0, 64));
  CK(hipMalloc(&y, (size_t)M / 2 * Cout * 4)); CK(hipMalloc(&yb, (size_t)M / 2 * Cout * 2)); CK(hipMalloc(&idx, (size_t)M / 2 * Cout)); CK(hipMalloc(&bias, Cout * 4));
  CK(hipMemset(bias, 0, Cout * 4));
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, x, (size_t)B * H * W * Cin, 1u);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, w, (size_t)Cout * K, 2u);
  CK(hipDeviceSynchronize()); printf("filled\n");
  LoadConvKh a; a.src = x; a.g = g;
  LoadKh b; b.p = w; b.ld = K; b.rows = Cout; b.K = K;
  EpConv ep; ep.y = y; ep.idx = idx; ep.bias = bias; ep.Cout = Cout; ep.rows = M; ep.pmode = pool; ep.relu = 1; ep.yb = yb;
  printf("conv6 forward as implicit GEMM: M %d N %d K %d, %d workgroups\n", M, N, K, (N / 256) * ((M + 255) / 256));
  run<0>(a, b, ep, M, N, K, zero, "full kernel");
  run<1>(a, b, ep, M, N, K, zero, "no in-loop DMA (reads + MFMA + barrier)");
  // run<2>: faults (under investigation)
  run<6>(a, b, ep, M, N, K, zero, "DMA + barrier only");
  run<3>(a, b, ep, M, N, K, zero, "reads + barrier only");
  run<5>(a, b, ep, M, N, K, zero, "MFMA + barrier only");
  run<7>(a, b, ep, M, N, K, zero, "barrier + epilogue only");
  { EpConv e2 = ep; e2.y = nullptr; run<7>(a, b, e2, M, N, K, zero, "barrier + epilogue only, no fp32 y");
    run<0, false>(a, b, e2, M, N, K, zero, "full kernel, plain reads, no fp32 y");
    e2.idx = nullptr; run<7>(a, b, e2, M, N, K, zero, "barrier + epilogue only, bf16 y only");
    e2.yb = nullptr; run<7>(a, b, e2, M, N, K, zero, "barrier + epilogue only, no stores at all"); }
  for (int rep = 0; rep < 2; ++rep) {                     // round 2: how much of the launch is the im2col (A) stream?
    run<0, false>(a, b, ep, M, N, K, zero, "full kernel, plain reads");
    run<16, false>(a, b, ep, M, N, K, zero, "plain reads, A pieces from the zero page");
    run<32, false>(a, b, ep, M, N, K, zero, "plain reads, no A pieces");
    run<128, false>(a, b, ep, M, N, K, zero, "plain reads, B pieces from the zero page");
    run<128 + 16, false>(a, b, ep, M, N, K, zero, "plain reads, A and B pieces from the zero page");
  }
  for (int rep = 0; rep < 1; ++rep) {
    run<0, true>(a, b, ep, M, N, K, zero, "full kernel, pipelined reads");
    run<0, false>(a, b, ep, M, N, K, zero, "full kernel, plain reads");
    run<0, false, true>(a, b, ep, M, N, K, zero, "full kernel, two tiles per barrier");
    run<8, false>(a, b, ep, M, N, K, zero, "full kernel, s_setprio around the MFMA blocks");
  }
  return 0;
}
