// ops_gemm.hip -- instantiations + launchers of the MFMA contractions: plain GEMMs (nn.Linear forward /
// gradInput / gradWeight, LSTM.lua:79-88, model_utils.lua:57-116), implicit-GEMM convolutions
// (cudnn.SpatialConvolution, cnn.lua:17-42) and the fused recurrent-step kernels.
#include "ops.h"

namespace aocr {

template <class AL, class BL, class EP>
static void launch_big(hipStream_t s, bool bf16, const AL& a, const BL& b, const EP& ep, int M, int N, int K, int ksplit) {
  if (M <= 0 || N <= 0) return;
  const int chunk = bf16 ? 32 : 8;
  if (ksplit < 1) ksplit = 1;
  int kper = cdiv(cdiv(K, ksplit), chunk) * chunk;
  if (kper < chunk) kper = chunk;
  ksplit = cdiv(K, kper); if (ksplit < 1) ksplit = 1;
  const int gx = cdiv(N, 128), gy = cdiv(M, 128);
  if (bf16) hipLaunchKernelGGL((gemm_lds_bf16_kernel<AL, BL, EP>), dim3(gx * gy, 1, ksplit), dim3(256), 0, s, a, b, ep, K, kper, gx, gy);
  else      hipLaunchKernelGGL((gemm_big_kernel<false, 2, 2, AL, BL, EP>), dim3(gx, gy, ksplit), dim3(256), 0, s, a, b, ep, K, kper);
}

void launch_big_kk(hipStream_t s, bool bf16, const LoadK& a, const LoadK& b, const EpStore& ep, int M, int N, int K, int ksplit) {
  launch_big(s, bf16, a, b, ep, M, N, K, ksplit);
}
void launch_big_kmn(hipStream_t s, bool bf16, const LoadK& a, const LoadMN& b, const EpStore& ep, int M, int N, int K, int ksplit) {
  launch_big(s, bf16, a, b, ep, M, N, K, ksplit);
}
void launch_big_mnmn(hipStream_t s, bool bf16, const LoadMN& a, const LoadMN& b, const EpStore& ep, int M, int N, int K, int ksplit) {
  launch_big(s, bf16, a, b, ep, M, N, K, ksplit);
}
void launch_conv_fwd(hipStream_t s, bool bf16, const LoadConvK& a, const LoadK& b, const EpConv& ep, int M, int N, int K) {
  launch_big(s, bf16, a, b, ep, M, N, K, 1);
}
void launch_conv_dgrad(hipStream_t s, bool bf16, const LoadConvK& a, const LoadConvWT& b, const EpStore& ep, int M, int N, int K) {
  launch_big(s, bf16, a, b, ep, M, N, K, 1);
}
void launch_conv_wgrad(hipStream_t s, bool bf16, const LoadMN& a, const LoadConvXcol& b, const EpStore& ep, int M, int N, int K, int ksplit) {
  launch_big(s, bf16, a, b, ep, M, N, K, ksplit);
}

template <int NT, bool GATES, class ARGS>
static void launch_small(hipStream_t s, bool bf16, int nz, const ARGS* z, int M, int ncols, int gate_stride) {
  if (M <= 0 || ncols <= 0) return;
  dim3 grid(GATES ? cdiv(ncols, 32) : cdiv(ncols, 32 * NT), cdiv(M, 32), nz);
  const ARGS& z0 = z[0]; const ARGS& z1 = z[nz > 1 ? 1 : 0];
  if (bf16) hipLaunchKernelGGL((gemm_small_kernel<true, NT, GATES, decltype(z0.a), decltype(z0.b), decltype(z0.ep)>), grid,
                               dim3(256), 0, s, z0, z1, gate_stride);
  else      hipLaunchKernelGGL((gemm_small_kernel<false, NT, GATES, decltype(z0.a), decltype(z0.b), decltype(z0.ep)>), grid,
                               dim3(256), 0, s, z0, z1, gate_stride);
}
void launch_small_gates_fwd(hipStream_t s, bool bf16, int nz, const GatesFwdArgs* z, int M, int H) {
  launch_small<4, true>(s, bf16, nz, z, M, H, H);
}
void launch_small_kk(hipStream_t s, bool bf16, int nz, const SmallKKArgs* z, int M, int N) {
  launch_small<1, false>(s, bf16, nz, z, M, N, 0);
}
void launch_small_kmn(hipStream_t s, bool bf16, int nz, const SmallKMNArgs* z, int M, int N) {
  launch_small<1, false>(s, bf16, nz, z, M, N, 0);
}
void launch_small_gates_bwd(hipStream_t s, bool bf16, int nz, const GatesBwdArgs* z, int M, int H) {
  launch_small<1, false>(s, bf16, nz, z, M, H, 0);
}

// number of K slices so that a launch has >= ~768 workgroups (256 CUs x 3) when accumulation is atomic
static int pick_ksplit(int M, int N, int K, bool bf16) {
  int64_t blocks = (int64_t)cdiv(M, 128) * cdiv(N, 128);
  int chunk = bf16 ? 16 : 8;
  int ks = (int)((768 + blocks - 1) / blocks);
  int maxks = K / (chunk * 16); if (maxks < 1) maxks = 1;
  if (ks > maxks) ks = maxks;
  if (ks < 1) ks = 1;
  return ks;
}

int gemm(hipStream_t s, bool bf16, const float* A, int64_t lda, bool a_k, const float* B, int64_t ldb, bool b_k, float* C,
          int64_t ldc, int M, int N, int K, const float* bias, const float* bias2, int flags) {
  EpStore ep = make_store(C, ldc, M, N, bias, bias2, flags);
  int ks = (flags & EP_ATOMIC) ? pick_ksplit(M, N, K, bf16) : 1;
  if (a_k && b_k) launch_big_kk(s, bf16, make_loadk(A, lda, M, K), make_loadk(B, ldb, N, K), ep, M, N, K, ks);
  else if (a_k && !b_k) launch_big_kmn(s, bf16, make_loadk(A, lda, M, K), make_loadmn(B, ldb, N, K), ep, M, N, K, ks);
  else if (!a_k && !b_k) launch_big_mnmn(s, bf16, make_loadmn(A, lda, M, K), make_loadmn(B, ldb, N, K), ep, M, N, K, ks);
  else return -1;   // A^T * B^T never occurs on the hot path
  return 0;
}

// ---------------------------------------------------------------------------------------------
// convolution layers, channels-last.  Output grid Ho = H + 2*pad - ks + 1 (stride 1).
// ---------------------------------------------------------------------------------------------
void conv_forward(hipStream_t s, bool bf16, const float* x, const float* w, const float* bias, float* y, uint8_t* idx, int B,
                  int H, int W, int Cin, int Cout, int ks, int pad, int relu, int pool) {
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  LoadConvK a; a.src = x; a.H = H; a.W = W; a.C = Cin; a.KW = ks; a.sgn = 1; a.off = -pad; a.Hr = Ho; a.Wr = Wo;
  a.pmode = pool; a.Hp = Ho / 2; a.Wp = Wo / 2;
  if (pool == 1) a.rows = B * a.Hp * a.Wp * 4; else if (pool == 2) a.rows = B * a.Hp * Wo * 2; else a.rows = B * Ho * Wo;
  a.K = ks * ks * Cin;
  LoadK b = make_loadk(w, a.K, Cout, a.K);
  EpConv ep; ep.y = y; ep.idx = idx; ep.bias = bias; ep.Cout = Cout; ep.rows = a.rows; ep.pmode = pool; ep.relu = relu;
  launch_conv_fwd(s, bf16, a, b, ep, a.rows, Cout, a.K);
}

void conv_backward_data(hipStream_t s, bool bf16, const float* dy, const float* w, float* dx, int B, int H, int W, int Cin,
                        int Cout, int ks, int pad) {
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  LoadConvK a; a.src = dy; a.H = Ho; a.W = Wo; a.C = Cout; a.KW = ks; a.sgn = -1; a.off = pad; a.Hr = H; a.Wr = W;
  a.pmode = 0; a.Hp = 0; a.Wp = 0; a.rows = B * H * W; a.K = ks * ks * Cout;
  LoadConvWT b; b.w = w; b.Cin = Cin; b.Cout = Cout; b.KK = ks * ks; b.K = a.K;
  EpStore ep = make_store(dx, Cin, a.rows, Cin);
  launch_conv_dgrad(s, bf16, a, b, ep, a.rows, Cin, a.K);
}

void conv_backward_filter(hipStream_t s, bool bf16, const float* x, const float* dy, float* dw, float* dbias, int B, int H,
                          int W, int Cin, int Cout, int ks, int pad) {
  const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
  const int P = B * Ho * Wo, N = ks * ks * Cin;
  LoadMN a = make_loadmn(dy, Cout, Cout, P);
  LoadConvXcol b; b.x = x; b.H = H; b.W = W; b.Cin = Cin; b.KW = ks; b.pad = pad; b.Ho = Ho; b.Wo = Wo; b.N = N; b.K = P;
  EpStore ep = make_store(dw, N, Cout, N, nullptr, nullptr, EP_ATOMIC);
  launch_conv_wgrad(s, bf16, a, b, ep, Cout, N, P, pick_ksplit(Cout, N, P, bf16));
  if (dbias) colsum_accum(s, dy, Cout, P, Cout, dbias);
}

}  // namespace aocr
