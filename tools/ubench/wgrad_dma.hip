// Timing-only ablations of conv_wgrad_dma_kernel at the conv5 / conv6 filter-gradient shapes of workload C3
// (B=256, 4 x 64 map, 256|512 -> 512, 3x3): which of {LDS-DMA stream, transposed fragment reads, MFMA issue} bounds the K loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch-attention-ocr_amd/csrc wgrad_dma.hip -o wgrad_dma
#include <hip/hip_runtime.h>
#include <cstdio>
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

template <int ABL> static int run(const LoadMNh& a, const LoadConvXcolh& b, const EpStore& ep, int P, int N, int Cout, const bf16_t* zero, const char* name) {
  const int tiles = (N / 256) * (Cout / 256); int ks = 256 / tiles; int kper = ((P + ks - 1) / ks + 31) / 32 * 32; ks = (P + kper - 1) / kper;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((conv_wgrad_dma_kernel<EpStore, ABL>), dim3(tiles * ks), dim3(512), 0, 0, a, b, ep, P, kper, N / 256, Cout / 256, zero, ks, (float*)nullptr, 0ll);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((conv_wgrad_dma_kernel<EpStore, ABL>), dim3(tiles * ks), dim3(512), 0, 0, a, b, ep, P, kper, N / 256, Cout / 256, zero, ks, (float*)nullptr, 0ll);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / it;
  printf("%-52s %8.1f us  (%d workgroups x %d steps: %.2f us per step)\n", name, us, tiles * ks, kper / 32, us / (kper / 32));
  return 0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int B = 256, H = 4, W = 64, Cout = 512, ks = 3, pad = 1;
  for (int Cin : {256, 512}) {
    const int P = B * H * W, N = ks * ks * Cin;
    bf16_t *x, *dy, *zero; float* dw;
    CK(hipMalloc(&x, (size_t)P * Cin * 2)); CK(hipMalloc(&dy, (size_t)P * Cout * 2)); CK(hipMalloc(&zero, 64)); CK(hipMemset(zero, 0, 64));
    CK(hipMalloc(&dw, (size_t)Cout * N * 4)); CK(hipMemset(dw, 0, (size_t)Cout * N * 4));
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, x, (size_t)P * Cin, 1u);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, dy, (size_t)P * Cout, 2u);
    CK(hipDeviceSynchronize());
    LoadConvXcol g; g.x = nullptr; g.H = H; g.W = W; g.Cin = Cin; g.KW = ks; g.pad = pad; g.Ho = H; g.Wo = W; g.N = N; g.K = P;
    LoadMNh a; a.p = dy; a.ld = Cout; a.rows = Cout; a.K = P;
    LoadConvXcolh b; b.x = x; b.g = g;
    EpStore ep{}; ep.C = dw; ep.ldc = N; ep.M = Cout; ep.N = N; ep.bias = nullptr; ep.bias2 = nullptr; ep.flags = EP_ATOMIC; ep.C1 = nullptr; ep.ldc1 = 0; ep.N0 = 0;
    printf("filter gradient %d -> %d: Cout %d x N %d over %d pixels\n", Cin, Cout, Cout, N, P);
    run<0>(a, b, ep, P, N, Cout, zero, "full kernel");
    run<1>(a, b, ep, P, N, Cout, zero, "no in-loop DMA (reads + MFMA + barrier)");
    run<6>(a, b, ep, P, N, Cout, zero, "DMA + barrier only");
    run<3>(a, b, ep, P, N, Cout, zero, "reads + barrier only");
    run<5>(a, b, ep, P, N, Cout, zero, "MFMA + barrier only");
    run<7>(a, b, ep, P, N, Cout, zero, "barrier + epilogue only");
    run<16>(a, b, ep, P, N, Cout, zero, "d y pieces from the zero page");
    run<128>(a, b, ep, P, N, Cout, zero, "x pieces from the zero page");
    run<144>(a, b, ep, P, N, Cout, zero, "both from the zero page");
    run<0>(a, b, ep, P, N, Cout, zero, "full kernel");
    { EpStore e2 = ep; e2.flags = 0; run<7>(a, b, e2, P, N, Cout, zero, "barrier + epilogue only, plain stores");
      run<0>(a, b, e2, P, N, Cout, zero, "full kernel, plain stores");
      LoadMNh a1 = a; LoadConvXcolh b1 = b; const int P1 = 32 * (256 / ((N / 256) * (Cout / 256)));
      run<7>(a1, b1, ep, P1, N, Cout, zero, "one step per workgroup: prologue + atomic epilogue");
      run<7>(a1, b1, e2, P1, N, Cout, zero, "one step per workgroup: prologue + plain-store epilogue"); }
    CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(dw)); CK(hipFree(zero));
  }
  return 0;
}
