// conv_wgrad_halo_kernel (round 4): correctness against a direct evaluation on a small problem, then timing at the conv4 / conv5 / conv6 filter-gradient
// shapes of workload C3 next to conv_wgrad_dma_kernel (both with split-K slabs + splitk sum left out: kernel time only).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch-attention-ocr_amd/csrc wgrad_halo.hip -o wgrad_halo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
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
// direct evaluation: one thread per (co, tap, ci)
__global__ void ref_kernel(const bf16_t* dy, const bf16_t* x, float* dw, int B, int H, int W, int Cin, int Cout) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, N = 9 * Cin;
  if (idx >= Cout * N) return;
  const int co = idx / N, n = idx - co * N, tap = n / Cin, ci = n - tap * Cin, ky = tap / 3 - 1, kx = tap % 3 - 1;
  float s = 0.f;
  for (int b = 0; b < B; ++b) for (int y = 0; y < H; ++y) for (int xx = 0; xx < W; ++xx) {
    const int sy = y + ky, sx = xx + kx;
    if ((unsigned)sy >= (unsigned)H || (unsigned)sx >= (unsigned)W) continue;
    s += (float)dy[((size_t)(b * H + y) * W + xx) * Cout + co] * (float)x[((size_t)(b * H + sy) * W + sx) * Cin + ci];
  }
  dw[idx] = s;
}
__global__ void sum_slabs(const float* part, int ks, size_t n, float* out) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float s = 0.f; for (int z = 0; z < ks; ++z) s += part[z * n + i]; out[i] = s; }
}

static int launch_halo(const bf16_t* dy, const bf16_t* x, float* part, int B, int H, int W, int Cin, int Cout, const bf16_t* zero, int& ks_out) {
  const int mt = Cout % 256 == 0 ? 256 : 128;
  const int gx = Cin / 32, gy = Cout / mt, tiles = gx * gy, S = B * H * (W / 32);
  int ks = tiles >= 256 ? 1 : 256 / tiles; if (ks > S) ks = S;
  const int per = (S + ks - 1) / ks; ks = (S + per - 1) / per;
  if (mt == 256) hipLaunchKernelGGL((conv_wgrad_halo_kernel<0, 4>), dim3(tiles * ks), dim3(512), 0, 0, dy, x, part, (long long)Cout * 9 * Cin, B, H, W, Cin, Cout, gx, gy, ks, per, zero);
  else hipLaunchKernelGGL((conv_wgrad_halo_kernel<0, 2>), dim3(tiles * ks), dim3(256), 0, 0, dy, x, part, (long long)Cout * 9 * Cin, B, H, W, Cin, Cout, gx, gy, ks, per, zero);
  ks_out = ks;
  return 0;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int warm_launches = argc > 1 ? atoi(argv[1]) : 3;          // e.g. 6000 launches of ~0.25 ms = 1.5 s of sustained load before the timed 20
  printf("warm-up launches before every timed loop: %d\n", warm_launches);
  bf16_t* zero; CK(hipMalloc(&zero, 64)); CK(hipMemset(zero, 0, 64));
  for (int Cout : {256, 128}) {   // ---- correctness: two images of 5 x 64, 64 -> 256 (eight waves) and 64 -> 128 (four waves)
    const int B = 2, H = 5, W = 64, Cin = 64, P = B * H * W, N = 9 * Cin;
    bf16_t *x, *dy; float *part, *dw, *ref;
    CK(hipMalloc(&x, (size_t)P * Cin * 2)); CK(hipMalloc(&dy, (size_t)P * Cout * 2)); CK(hipMalloc(&part, (size_t)64 * Cout * N * 4)); CK(hipMalloc(&dw, (size_t)Cout * N * 4)); CK(hipMalloc(&ref, (size_t)Cout * N * 4));
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, x, (size_t)P * Cin, 1u); hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, 0, dy, (size_t)P * Cout, 2u);
    int ks = 0; launch_halo(dy, x, part, B, H, W, Cin, Cout, zero, ks);
    hipLaunchKernelGGL(sum_slabs, dim3(256), dim3(256), 0, 0, part, ks, (size_t)Cout * N, dw);
    hipLaunchKernelGGL(ref_kernel, dim3((Cout * N + 255) / 256), dim3(256), 0, 0, dy, x, ref, B, H, W, Cin, Cout);
    CK(hipDeviceSynchronize());
    std::vector<float> a((size_t)Cout * N), b((size_t)Cout * N);
    CK(hipMemcpy(a.data(), dw, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref, b.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0; size_t worst = 0;
    for (size_t i = 0; i < a.size(); ++i) { const double d = fabs((double)a[i] - b[i]); if (d > md) { md = d; worst = i; } mx = fmax(mx, fabs((double)b[i])); }
    printf("correctness (B %d, %d x %d, %d -> %d, split-K %d): max abs diff %.3e of max |dW| %.3e at (co %zu, n %zu)\n", B, H, W, Cin, Cout, ks, md, mx, worst / N, worst % N);
    CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(part)); CK(hipFree(dw)); CK(hipFree(ref));
  }
  const int B = 256;
  struct Shape { int H, W, Cin, Cout; const char* name; } shapes[] = {{16, 128, 64, 128, "conv2"}, {8, 64, 128, 256, "conv3"}, {8, 64, 256, 256, "conv4"}, {4, 64, 256, 512, "conv5"}, {4, 64, 512, 512, "conv6"}};
  for (const Shape& sh : shapes) {
    const int H = sh.H, W = sh.W, Cin = sh.Cin, Cout = sh.Cout, P = B * H * W, N = 9 * Cin;
    bf16_t *x, *dy; float *part, *dw;
    CK(hipMalloc(&x, (size_t)P * Cin * 2)); CK(hipMalloc(&dy, (size_t)P * Cout * 2)); CK(hipMalloc(&part, (size_t)64 << 22)); CK(hipMalloc(&dw, (size_t)Cout * N * 4));
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, x, (size_t)P * Cin, 1u); hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, dy, (size_t)P * Cout, 2u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int ks = 0;
    for (int i = 0; i < warm_launches; ++i) launch_halo(dy, x, part, B, H, W, Cin, Cout, zero, ks);      // steady state: the chip's clock under SUSTAINED load (MI355X_MICROARCH.md, DVFS give-back (6))
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    const int it = 20;
    for (int i = 0; i < it; ++i) launch_halo(dy, x, part, B, H, W, Cin, Cout, zero, ks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / it, gf = 2.0 * P * Cout * N / 1e9;
    printf("%s filter gradient (%d -> %d, %d x %d, %.0f GFLOP): halo-resident kernel %8.1f us = %6.1f TFLOP/s (split-K %d, %d workgroups)\n", sh.name, Cin, Cout, H, W, gf, us, gf / us * 1e3, ks, (Cin / 32) * (Cout / 256) * ks);
    // the round-3 kernel at the same shape, slab stores
    if (Cout % 256 == 0 && N % 256 == 0 && N >= 2304) {
      LoadConvXcol g; g.x = nullptr; g.H = H; g.W = W; g.Cin = Cin; g.KW = 3; g.pad = 1; g.Ho = H; g.Wo = W; g.N = N; g.K = P;
      LoadMNh a; a.p = dy; a.ld = Cout; a.rows = Cout; a.K = P;
      LoadConvXcolh b; b.x = x; b.g = g;
      EpStore ep{}; ep.C = dw; ep.ldc = N; ep.M = Cout; ep.N = N; ep.flags = EP_ATOMIC;
      const int tiles = (N / 256) * (Cout / 256); int ks2 = 256 / tiles; int kper = ((P + ks2 - 1) / ks2 + 31) / 32 * 32; ks2 = (P + kper - 1) / kper;
      for (int i = 0; i < warm_launches; ++i) hipLaunchKernelGGL((conv_wgrad_dma_kernel<EpStore, 0>), dim3(tiles * ks2), dim3(512), 0, 0, a, b, ep, P, kper, N / 256, Cout / 256, zero, ks2, part, (long long)Cout * N);
      CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      for (int i = 0; i < it; ++i) hipLaunchKernelGGL((conv_wgrad_dma_kernel<EpStore, 0>), dim3(tiles * ks2), dim3(512), 0, 0, a, b, ep, P, kper, N / 256, Cout / 256, zero, ks2, part, (long long)Cout * N);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us2 = ms * 1e3 / it;
      printf("%s                                                      conv_wgrad_dma_kernel %8.1f us = %6.1f TFLOP/s (split-K %d)\n", sh.name, us2, gf / us2 * 1e3, ks2);
    }
    CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(part)); CK(hipFree(dw));
  }
  return 0;
}
