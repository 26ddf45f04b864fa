// Sustained v_mfma_f32_32x32x2_f32 rate of the chip (no memory traffic): what "the fp32 MFMA peak" is worth once the clock has settled.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip && ./mfma_f32_rate
// Prints TFLOP/s for 1, 2 and 4 waves per SIMD and 1-4 independent accumulators per wave, cold (first launch) and after a 0.5 s warm-up.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC> __global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float x, float y) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = x + threadIdx.x * 1e-6f, b = y;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.678f) out[0] = s;
}
template <int NACC> static double run(int wgs, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<NACC>), dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)wgs * 4 * iters * 16 * NACC * 4096.0;
  return flop / (ms * 1e-3) * 1e-12;
}
int main() {
  float* d; hipMalloc(&d, 4);
  const int cfg[3] = {256, 512, 1024};          // workgroups of 4 waves: 1, 2, 4 waves per SIMD on 256 CUs
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) { for (int k = 0; k < 12; ++k) run<4>(1024, 20000, d); }      // ~0.5 s of MFMA: the clock has settled
    for (int c = 0; c < 3; ++c) {
      const int it = 40000 / (cfg[c] / 256);
      printf("%s  %4d workgroups (%d wave%s per SIMD): 1 acc %6.1f  2 acc %6.1f  4 acc %6.1f TFLOP/s\n", pass ? "warm" : "cold", cfg[c], cfg[c] / 256, cfg[c] > 256 ? "s" : " ",
             run<1>(cfg[c], it, d), run<2>(cfg[c], it / 2, d), run<4>(cfg[c], it / 4, d));
    }
  }
  // 200 of 256 CUs busy (the C2 grids): per-CU rate
  printf("warm   200 workgroups (1 wave per SIMD on 200 CUs): 4 acc %6.1f TFLOP/s\n", run<4>(200, 10000, d));
  return 0;
}
