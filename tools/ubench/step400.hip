// Experiment (round 6): the gate product of one decoder step at the reference's default shape -- M = 400 rows, Hd = 1024 (N = 4 x 1024 gate columns),
// K = 2048 = [x | h_prev] x [W_i2h | W_h2h] -- through the library's own step kernels and epilogue, alternating between two weight sets the way a step alternates
// between its two layers (40 MB of weights per step cycle through the 4 MB L2 of every XCD).  Questions: (1) does the 2 KB row stride of the operands cost L2 channel
// parallelism (ld = 1024 against ld = 1024 + 64 elements); (2) what do other tile shapes of the same kernel buy; (3) candidates of a large-M step kernel.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch-attention-ocr_amd/csrc step400.hip -o step400
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "ops.h"
#include "mfma_gemm.h"
#include "stepl.h"
using namespace aocr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void fill(bf16_t* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)((((h & 0xffff) / 32768.0f) - 1.0f) * scale);
  }
}
__global__ void fillf(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (((h & 0xffff) / 32768.0f) - 1.0f) * scale;
  }
}

struct Bufs {
  int M, H, ld;
  bf16_t *x, *hp, *w[2][2];                 // two weight sets x {W_i2h, W_h2h}, each [4H][ld]
  float *zx, *cprev, *c, *h, *h2, *gates; bf16_t *hb, *hb2;
};
static Bufs make(int M, int H, int ld) {
  Bufs b; b.M = M; b.H = H; b.ld = ld;
  CK(hipMalloc(&b.x, (size_t)M * ld * 2)); CK(hipMalloc(&b.hp, (size_t)M * ld * 2));
  fill<<<512, 256>>>(b.x, (size_t)M * ld, 1, 1.f); fill<<<512, 256>>>(b.hp, (size_t)M * ld, 2, 1.f);
  for (int s = 0; s < 2; ++s) for (int k = 0; k < 2; ++k) { CK(hipMalloc(&b.w[s][k], (size_t)4 * H * ld * 2)); fill<<<1024, 256>>>(b.w[s][k], (size_t)4 * H * ld, 10 + 2 * s + k, 0.03f); }
  CK(hipMalloc(&b.zx, (size_t)M * 4 * H * 4)); fillf<<<512, 256>>>(b.zx, (size_t)M * 4 * H, 77, 0.5f);
  CK(hipMalloc(&b.cprev, (size_t)M * H * 4)); fillf<<<512, 256>>>(b.cprev, (size_t)M * H, 78, 1.f);
  CK(hipMalloc(&b.c, (size_t)M * H * 4)); CK(hipMalloc(&b.h, (size_t)M * H * 4)); CK(hipMalloc(&b.h2, (size_t)(M + 1) * 2 * H * 4)); CK(hipMalloc(&b.gates, (size_t)M * 4 * H * 4));
  CK(hipMalloc(&b.hb, (size_t)M * H * 2)); CK(hipMalloc(&b.hb2, (size_t)M * 2 * H * 2));
  CK(hipDeviceSynchronize());
  return b;
}
static EpGatesFwd epilogue(const Bufs& b) {
  EpGatesFwd e; e.zx = b.zx; e.ldzx = 4 * b.H; e.b1 = nullptr; e.b2 = nullptr; e.c_prev = b.cprev; e.ldcp = b.H; e.c_out = b.c; e.ldc = b.H; e.h_out = b.h; e.ldh = b.H;
  e.h_out2 = b.h2 + b.H; e.ldh2 = 2 * b.H; e.gates = b.gates; e.ldg = 4 * b.H; e.M = b.M; e.H = b.H; e.hb = b.hb; e.ldhb = b.H; e.hb2 = b.hb2 + b.H; e.ldhb2 = 2 * b.H;
  return e;
}
typedef SmallArgs2<LoadKh2, LoadKh2, EpGatesFwd> Z;
static bool g_cell4 = true;      // false: h_out2 off 16-byte alignment, so the kernel takes the one-unit-per-thread epilogue
static int g_epmode = 0;          // 0 full epilogue, 1 no epilogue at all (M = 0: every row dropped), 2 no saved gates / second copies / shadows
static Z args(const Bufs& b, int set) {
  Z zz; SmallArgs<LoadKh2, LoadKh2, EpGatesFwd> z;
  z.a = make_loadkh2(b.x, b.ld, b.H, b.hp, b.ld, b.H, b.M);
  z.b = make_loadkh2(b.w[set][0], b.ld, b.H, b.w[set][1], b.ld, b.H, 4 * b.H);
  z.ep = epilogue(b); z.K = 2 * b.H;
  if (!g_cell4) { z.ep.h_out2 = b.h2 + b.H + 1; z.ep.hb2 = nullptr; }
  if (g_epmode == 1) z.ep.M = 0;
  if (g_epmode == 2) { z.ep.gates = nullptr; z.ep.h_out2 = nullptr; z.ep.hb = nullptr; z.ep.hb2 = nullptr; }
  zz.z[0] = zz.z[1] = zz.z[2] = z; return zz;
}
template <class F> static float time_us(F&& launch, int iters = 40) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) launch(i & 1);
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch(i & 1);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1000.f / iters);
  }
  std::sort(t.begin(), t.end());
  return t[2];
}
static double checksum(const Bufs& b) {
  std::vector<float> h((size_t)b.M * b.H); CK(hipMemcpy(h.data(), b.h, h.size() * 4, hipMemcpyDeviceToHost));
  double s = 0; for (size_t i = 0; i < h.size(); ++i) s += (double)h[i] * (double)((i % 7) + 1);
  return s;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 400, H = argc > 2 ? atoi(argv[2]) : 1024;
  for (int mode : {0}) {
    g_epmode = mode; const int ld = H;
    Bufs b = make(M, H, ld);
    printf("--- M=%d H=%d K=%d ld=%d epilogue mode %d (0 full, 1 none, 2 c + h only)\n", M, H, 2 * H, ld, mode);
    { float us = time_us([&](int set) { Z zz = args(b, set); hipLaunchKernelGGL((gemm_step_kernel<4, 1, LoadKh2, EpGatesFwd, 4, 2>), dim3(H / 32, cdiv(M, 64), 1), dim3(256), 0, 0, zz, H); });
      printf("step<NT4,G1,NW4,MT2> grid %dx%d            : %7.2f us  (%.0f TF/s)  checksum %.6e\n", H / 32, cdiv(M, 64), us, 2.0 * M * 4 * H * 2 * H / us * 1e-6, checksum(b)); }
    { float us = time_us([&](int set) { Z zz = args(b, set); hipLaunchKernelGGL((gemm_step_kernel<4, 1, LoadKh2, EpGatesFwd, 4, 1>), dim3(H / 32, cdiv(M, 32), 1), dim3(256), 0, 0, zz, H); });
      printf("step<NT4,G1,NW4,MT1> grid %dx%d           : %7.2f us  checksum %.6e\n", H / 32, cdiv(M, 32), us, checksum(b)); }
    { float us = time_us([&](int set) { Z zz = args(b, set); hipLaunchKernelGGL((gemm_step_kernel<2, 2, LoadKh2, EpGatesFwd, 4, 4>), dim3(H / 16, cdiv(M, 128), 1), dim3(256), 0, 0, zz, H); });
      printf("step<NT2,G2,NW4,MT4> grid %dx%d            : %7.2f us  checksum %.6e\n", H / 16, cdiv(M, 128), us, checksum(b)); }
    { float us = time_us([&](int set) { Z zz = args(b, set); hipLaunchKernelGGL((gemm_step_kernel<2, 2, LoadKh2, EpGatesFwd, 8, 1>), dim3(H / 16, cdiv(M, 32), 1), dim3(512), 0, 0, zz, H); });
      printf("step<NT2,G2,NW8,MT1> grid %dx%d (the library's half-tile form) : %7.2f us  checksum %.6e\n", H / 16, cdiv(M, 32), us, checksum(b)); }
    { float us = time_us([&](int set) { Z zz = args(b, set); hipLaunchKernelGGL((gemm_step_kernel<2, 2, LoadKh2, EpGatesFwd, 4, 2>), dim3(H / 16, cdiv(M, 64), 1), dim3(256), 0, 0, zz, H); });
      printf("step<NT2,G2,NW4,MT2> grid %dx%d            : %7.2f us  checksum %.6e\n", H / 16, cdiv(M, 64), us, checksum(b)); }
#define STEPL(MT, NT, G, NS, SUBS, NW, SPLITN, GX, GY) { float us = time_us([&](int set) { Z zz = args(b, set); hipLaunchKernelGGL((gemm_stepl_kernel<MT, NT, G, EpGatesFwd, NS, SUBS, NW, SPLITN>), dim3((GX) * (GY)), dim3(64 * NW), 0, 0, zz, H, GX, GY); }); \
      printf("stepl<MT%d,NT%d,G%d,NS%d,SUBS%d,NW%d,SPLITN%d> grid %dx%d : %7.2f us  (%.0f TF/s)  checksum %.6e\n", MT, NT, G, NS, SUBS, NW, (int)SPLITN, GX, GY, us, 2.0 * M * 4 * H * 2 * H / us * 1e-6, checksum(b)); }
    STEPL(2, 4, 1, 6, 2, 8, true, H / 32, cdiv(M, 64))
    STEPL(2, 4, 1, 6, 2, 4, true, H / 32, cdiv(M, 64))
    g_cell4 = false;
    STEPL(2, 4, 1, 6, 2, 8, true, H / 32, cdiv(M, 64))
    g_cell4 = true;
  }
  // ---- plain products of the backward step: C[M][N] = A[M][K] . W^T (W^T stored [N][K]), N = Hd, K = 4 Hd (d z W_h2h) and K = Hd, one and three problems per launch
  for (int KK : {4 * H, H}) {
    const int N = H;
    bf16_t *A[3], *W[3]; float* C[3]; bf16_t* Cb[3];
    for (int i = 0; i < 3; ++i) {
      CK(hipMalloc(&A[i], (size_t)M * KK * 2)); CK(hipMalloc(&W[i], (size_t)N * KK * 2)); CK(hipMalloc(&C[i], (size_t)M * N * 4)); CK(hipMalloc(&Cb[i], (size_t)M * N * 2));
      fill<<<512, 256>>>(A[i], (size_t)M * KK, 31 + i, 1.f); fill<<<512, 256>>>(W[i], (size_t)N * KK, 41 + i, 0.03f);
    }
    CK(hipDeviceSynchronize());
    typedef SmallArgs2<LoadKh2, LoadKh2, EpStore> ZS;
    auto pargs = [&]() { ZS zz; for (int i = 0; i < 3; ++i) { zz.z[i].a = make_loadkh(A[i], KK, M, KK); zz.z[i].b = make_loadkh(W[i], KK, N, KK); zz.z[i].ep = make_store(C[i], N, M, N, nullptr, nullptr, 0); zz.z[i].ep.Cb = Cb[i]; zz.z[i].ep.ldcb = N; zz.z[i].K = KK; } return zz; };
    auto csum = [&]() { std::vector<float> hh((size_t)M * N); CK(hipMemcpy(hh.data(), C[2], hh.size() * 4, hipMemcpyDeviceToHost)); double q = 0; for (size_t i = 0; i < hh.size(); ++i) q += (double)hh[i] * (double)((i % 7) + 1); return q; };
    for (int nz : {1, 3}) {
      printf("--- plain M=%d N=%d K=%d, %d problem(s) per launch\n", M, N, KK, nz);
      { float us = time_us([&](int) { ZS zz = pargs(); if (nz == 1) zz.z[0] = zz.z[2]; hipLaunchKernelGGL((gemm_step_kernel<1, 0, LoadKh2, EpStore, 8, 1>), dim3(N / 32, cdiv(M, 32), nz), dim3(512), 0, 0, zz, 0); });
        printf("step<NT1,G0,NW8,MT1> grid %dx%dx%d : %7.2f us  checksum %.6e\n", N / 32, cdiv(M, 32), nz, us, csum()); }
#define STEPP(MT, NT, NS, SUBS, NW, SPLITN) { const int GX = N / (32 * NT), GY = cdiv(M, 32 * MT); float us = time_us([&](int) { ZS zz = pargs(); if (nz == 1) zz.z[0] = zz.z[2]; hipLaunchKernelGGL((gemm_stepl_kernel<MT, NT, 0, EpStore, NS, SUBS, NW, SPLITN>), dim3(GX * GY * nz), dim3(64 * NW), 0, 0, zz, 0, GX, GY); }); \
        printf("stepl<MT%d,NT%d,G0,NS%d,SUBS%d,NW%d,SPLITN%d> grid %dx%dx%d : %7.2f us  (%.0f TF/s)  checksum %.6e\n", MT, NT, NS, SUBS, NW, (int)SPLITN, GX, GY, nz, us, 2.0 * nz * M * N * KK / us * 1e-6, csum()); }
      STEPP(2, 4, 6, 2, 8, true)
      STEPP(2, 2, 4, 2, 8, true)
      STEPP(4, 2, 6, 2, 8, false)
    }
    for (int i = 0; i < 3; ++i) { hipFree(A[i]); hipFree(W[i]); hipFree(C[i]); hipFree(Cb[i]); }
  }
  return 0;
}
