// aocr_cpu_ref.cc -- C++/OpenMP restatement of the reference's CPU path.
//
// TEST INFRASTRUCTURE ONLY.  Never linked into libaocr.so; only tests/, __graft_entry__ and bench.py's cpu_baseline leg load
// the shared object this file builds (oracle/cpu_ref/libaocr_cpu_ref.so, recipe: oracle/cpu_ref/Makefile).
//
// PARITY UNPINNED: the reference (da03/torch-Attention-OCR) is Lua/Torch7, ships no vectors and cannot run here.  This is a
// SECOND restatement, written independently of oracle/oracle_torch.py (which sits on PyTorch's conv / batch_norm / autograd):
// every operator is spelled out the way the Torch7 CPU packages the reference `require`s execute it [upstream] --
//   * convolution = unfold (im2col) + GEMM per image, as nn.SpatialConvolutionMM (what cudnn.convert(net, nn) leaves on the CPU,
//     SURVEY.md S4); gradInput = col2im(W^T gradOutput); gradWeight += gradOutput . unfolded^T;
//   * nn.SpatialMaxPooling with stored arg-max, nn.SpatialBatchNormalization (biased variance to normalise, unbiased into
//     running_var, momentum 0.1, eps 1e-5), ReLU as a mask on the output;
//   * one nn.Linear (addmm) per i2h / h2h per time step, gate order [in, forget, out, g] (LSTM.lua:79-105), no fusion;
//   * Luong attention as two batched matrix products around a SoftMax (LSTM.lua:124-162); LogSoftMax + ClassNLLCriterion with
//     weight[PAD] = 0, sizeAverage = false (output_projector.lua:3-8, criterion.lua:3-9);
//   * the forward / BPTT order of model.lua:284-316, 537-569, 634-694, the clip + update of optim_sgd.lua:38-95, and the
//     beam / gold decode of model.lua:321-627 (S9 fixed, S10 = sorted with ties to the lowest index, as DESIGN.md documents).
// Arithmetic type: template parameter (double = the reference's CPU tensors; float for the fp32 timing).
// Two restatements agreeing to ~1e-10 (tests/test_cpu_ref.py, against tests/golden/*.npz and the Python oracle) is the
// parity-narrowing evidence available without a Torch7 artefact.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <cstdio>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int PAD = 1, EOS = 3;      // train.lua:53 (1-based ids)

struct Cfg { int32_t img_h, enc_hidden, enc_layers, dec_layers, vocab, emb, input_feed; };

int nthreads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

inline int small_team(long items) { return (int)std::max<long>(1, std::min<long>(std::min<long>(items, nthreads()), 16)); }   // per-timestep element-wise loops

// ------------------------------------------------------------------------------------------------ GEMM
// C[M,N] (ldc) (+)= sum_k a(i,k) * B[k*ldb + j],  a(i,k) = A[i*sai + k*sak]; B and C have unit stride along j.
// 6 x (2 vectors of 64 bytes) register tile, k blocked by 256; `par` spreads (row block, column tile) tasks over the OpenMP team.
template <class T> struct Vec { typedef T type __attribute__((vector_size(64), aligned(sizeof(T)))); static constexpr int n = 64 / sizeof(T); };

template <class T, int MR, int NV>
__attribute__((always_inline)) inline void micro_full(int K, const T* A, long sai, long sak, const T* B, long ldb, T* C, long ldc, bool acc) {
  typedef typename Vec<T>::type vec; constexpr int VL = Vec<T>::n;
  vec c[MR][NV];
  for (int r = 0; r < MR; ++r) for (int v = 0; v < NV; ++v) {
    if (acc) c[r][v] = *reinterpret_cast<const vec*>(C + r * ldc + v * VL); else c[r][v] = vec{} ;
  }
  for (int k = 0; k < K; ++k) {
    vec b[NV];
    for (int v = 0; v < NV; ++v) b[v] = *reinterpret_cast<const vec*>(B + k * ldb + v * VL);
    for (int r = 0; r < MR; ++r) { const T a = A[r * sai + k * sak]; for (int v = 0; v < NV; ++v) c[r][v] += a * b[v]; }
  }
  for (int r = 0; r < MR; ++r) for (int v = 0; v < NV; ++v) *reinterpret_cast<vec*>(C + r * ldc + v * VL) = c[r][v];
}
template <class T>
__attribute__((always_inline)) inline void micro_edge(int mr, int nr, int K, const T* A, long sai, long sak, const T* B, long ldb, T* C, long ldc, bool acc) {
  for (int r = 0; r < mr; ++r) for (int j = 0; j < nr; ++j) {
    T s = acc ? C[r * ldc + j] : T(0);
    for (int k = 0; k < K; ++k) s += A[r * sai + k * sak] * B[k * ldb + j];
    C[r * ldc + j] = s;
  }
}
// The register-tile kernel is compiled three times (AVX-512 / AVX2+FMA / baseline) and picked once at load time by CPU feature (the
// build host and the GPU box's host differ); the OpenMP region lives in the caller.
constexpr int GEMM_MR = 6, GEMM_NV = 2;
#define AOCR_TILE(NAME, TGT, T)                                                                                   \
  __attribute__((target(TGT), noinline)) void NAME(int kc, const T* a, const T* b, T* C, long ldc, bool acc) {    \
    micro_full<T, GEMM_MR, GEMM_NV>(kc, a, 1, GEMM_MR, b, GEMM_NV * Vec<T>::n, C, ldc, acc);                      \
  }
AOCR_TILE(tile_f64_512, "avx512f,fma,prefer-vector-width=512", double)
AOCR_TILE(tile_f64_256, "avx2,fma", double)
AOCR_TILE(tile_f32_512, "avx512f,fma,prefer-vector-width=512", float)
AOCR_TILE(tile_f32_256, "avx2,fma", float)
void tile_f64_base(int kc, const double* a, const double* b, double* C, long ldc, bool acc) { micro_full<double, GEMM_MR, GEMM_NV>(kc, a, 1, GEMM_MR, b, GEMM_NV * Vec<double>::n, C, ldc, acc); }
void tile_f32_base(int kc, const float* a, const float* b, float* C, long ldc, bool acc) { micro_full<float, GEMM_MR, GEMM_NV>(kc, a, 1, GEMM_MR, b, GEMM_NV * Vec<float>::n, C, ldc, acc); }
typedef void (*tile64_fn)(int, const double*, const double*, double*, long, bool);
typedef void (*tile32_fn)(int, const float*, const float*, float*, long, bool);
int cpu_level() { __builtin_cpu_init(); return __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0; }
const int CPU_LEVEL = cpu_level();
const tile64_fn TILE64 = CPU_LEVEL == 2 ? tile_f64_512 : CPU_LEVEL == 1 ? tile_f64_256 : tile_f64_base;
const tile32_fn TILE32 = CPU_LEVEL == 2 ? tile_f32_512 : CPU_LEVEL == 1 ? tile_f32_256 : tile_f32_base;
inline void tile(int kc, const double* a, const double* b, double* C, long ldc, bool acc) { TILE64(kc, a, b, C, ldc, acc); }
inline void tile(int kc, const float* a, const float* b, float* C, long ldc, bool acc) { TILE32(kc, a, b, C, ldc, acc); }

// Both operands are packed before use ([kc][NR] and [kc][MR] contiguous): leading dimensions here are powers of two (H*W of a
// feature map, 4H), and an unpacked 32-column panel with a 1-4 KiB row stride lands in a handful of L1 sets.
template <class T> struct GemmArgs { int M, N, K; const T* A; long sai, sak; const T* B; long ldb; T* C; long ldc; bool acc; };
template <class T> struct GemmBlk { static constexpr int MR = GEMM_MR, NR = GEMM_NV * Vec<T>::n, KC = 256, IT = 16, IB = IT * MR, JT = 8, JB = JT * NR; };
// one (row block, column block) task; bpack / apack are the calling thread's scratch panels
template <class T> void gemm_task(const GemmArgs<T>& g, long tt, int jb, T* bpack, T* apack) {
  typedef GemmBlk<T> Q; constexpr int MR = Q::MR, NR = Q::NR, KC = Q::KC;
  const int bi = (int)(tt / jb), bj = (int)(tt % jb);
  const int i_beg = bi * Q::IB, i_end = std::min(g.M, i_beg + Q::IB), j_beg = bj * Q::JB, j_end = std::min(g.N, j_beg + Q::JB);
  for (int k0 = 0; k0 < g.K; k0 += KC) {
    const int kc = std::min(KC, g.K - k0);
    const bool a2 = g.acc || k0 > 0;
    for (int i0 = i_beg, it = 0; i0 < i_end; i0 += MR, ++it) {                  // A micro-panels [kc][MR] (zero-padded rows), reused by every column tile
      T* ap = apack + (size_t)it * KC * MR; const T* Ap = g.A + i0 * g.sai + k0 * g.sak; const int mr = std::min(MR, i_end - i0);
      for (int r = 0; r < mr; ++r) for (int k = 0; k < kc; ++k) ap[k * MR + r] = Ap[r * g.sai + k * g.sak];
      for (int r = mr; r < MR; ++r) for (int k = 0; k < kc; ++k) ap[k * MR + r] = T(0);
    }
    for (int j0 = j_beg; j0 < j_end; j0 += NR) {
      const int nr = std::min(NR, j_end - j0);
      for (int k = 0; k < kc; ++k) {
        const T* src = g.B + (long)(k0 + k) * g.ldb + j0;
        for (int j = 0; j < nr; ++j) bpack[k * NR + j] = src[j];
        for (int j = nr; j < NR; ++j) bpack[k * NR + j] = T(0);
      }
      for (int i0 = i_beg, it = 0; i0 < i_end; i0 += MR, ++it) {
        const int mr = std::min(MR, i_end - i0);
        T* Cp = g.C + (long)i0 * g.ldc + j0;
        if (mr == MR && nr == NR) tile(kc, apack + (size_t)it * KC * MR, bpack, Cp, g.ldc, a2);
        else {                                                                   // edge tile: full kernel on the zero-padded panels into a scratch tile
          alignas(64) T ct[MR * NR];
          for (int r = 0; r < MR; ++r) for (int j = 0; j < NR; ++j) ct[r * NR + j] = (a2 && r < mr && j < nr) ? Cp[r * g.ldc + j] : T(0);
          tile(kc, apack + (size_t)it * KC * MR, bpack, ct, NR, true);
          for (int r = 0; r < mr; ++r) for (int j = 0; j < nr; ++j) Cp[r * g.ldc + j] = ct[r * NR + j];
        }
      }
    }
  }
}
template <class T>
void gemm(int M, int N, int K, const T* A, long sai, long sak, const T* B, long ldb, T* C, long ldc, bool acc, bool par) {
  typedef GemmBlk<T> Q;
  if (M <= 0 || N <= 0) return;
  if (K <= 0) { if (!acc) for (int i = 0; i < M; ++i) std::fill(C + i * ldc, C + i * ldc + N, T(0)); return; }
  const GemmArgs<T> g{M, N, K, A, sai, sak, B, ldb, C, ldc, acc};
  const int ib = (M + Q::IB - 1) / Q::IB, jb = (N + Q::JB - 1) / Q::JB;
  const long tasks = (long)ib * jb;
  if (par && tasks > 1) {
    // team size by the amount of work: a 128-thread team costs more in fork / barrier time than a per-timestep GEMM takes
    const double mflop = 2.0 * M * N * (double)K / 1e6;
    const int team = (int)std::max<long>(1, std::min<long>(std::min<long>(tasks, nthreads()), std::min<long>(32, (long)(mflop / 2.0))));
#pragma omp parallel num_threads(team)
    {
      std::vector<T> bpack((size_t)Q::KC * Q::NR + 16), apack((size_t)Q::IT * Q::KC * Q::MR + 16);
#pragma omp for schedule(dynamic)
      for (long tt = 0; tt < tasks; ++tt) gemm_task(g, tt, jb, bpack.data(), apack.data());
    }
  } else {                                                 // serial (also the form used inside an outer parallel-over-images region)
    static thread_local std::vector<T> bpack((size_t)Q::KC * Q::NR + 16), apack((size_t)Q::IT * Q::KC * Q::MR + 16);
    for (long tt = 0; tt < tasks; ++tt) gemm_task(g, tt, jb, bpack.data(), apack.data());
  }
}


// nn.Linear forward: Y[M,N] = X[M,K] W[N,K]^T (+ b) -- addmm against the transposed weight (transposed once per call)
template <class T> struct Linear {
  std::vector<T> wt; int N = 0, K = 0;
  void prepare(const T* W, int N_, int K_) {
    N = N_; K = K_; wt.resize((size_t)N * K);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; ++k) for (int n = 0; n < N; ++n) wt[(size_t)k * N + n] = W[(size_t)n * K + k];
  }
  // Y (+)= X W^T ; X row stride ldx
  void fwd(const T* X, long ldx, int M, T* Y, long ldy, bool acc) const { gemm(M, N, K, X, ldx, 1, wt.data(), N, Y, ldy, acc, true); }
};
// gradInput of nn.Linear: dX[M,K] (+)= dY[M,N] W[N,K]
template <class T> void linear_bwd_input(const T* dY, long lddy, const T* W, int M, int N, int K, T* dX, long lddx, bool acc) {
  gemm(M, K, N, dY, lddy, 1, W, K, dX, lddx, acc, true);
}
// gradWeight of nn.Linear: dW[N,K] += dY[M,N]^T X[M,K]
template <class T> void linear_bwd_weight(const T* dY, long lddy, const T* X, long ldx, int M, int N, int K, T* dW) {
  gemm(N, K, M, dY, 1, lddy, X, ldx, dW, K, true, true);
}
template <class T> void colsum_add(const T* dY, long ld, int M, int N, T* db) {
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) db[n] += dY[(size_t)m * ld + n];
}

// ------------------------------------------------------------------------------------------------ parameter views (oracle param_spec order)
struct ConvSpec { int idx, cin, cout, k, pad; };
const ConvSpec CONVS[7] = {{1, 1, 64, 3, 1}, {2, 64, 128, 3, 1}, {3, 128, 256, 3, 1}, {4, 256, 256, 3, 1}, {5, 256, 512, 3, 1}, {6, 512, 512, 3, 1}, {7, 512, 512, 2, 0}};

template <class T> struct LstmW { T *wi, *bi, *wh, *bh; int in, H; };
template <class T> struct Params {
  T *cw[8], *cb[8], *bnw[8], *bnb[8];
  std::vector<LstmW<T>> enc[2], dec;
  T *lookup, *wa, *wc, *wo, *bo;
  size_t total;
  void bind(T* p, const Cfg& c) {          // Torch7 getParameters() order: cnn.lua:9-45, model.lua:103-106,150
    size_t o = 0;
    for (int i = 0; i < 7; ++i) {
      const ConvSpec& s = CONVS[i];
      cw[s.idx] = p + o; o += (size_t)s.cout * s.cin * s.k * s.k; cb[s.idx] = p + o; o += s.cout;
      if (s.idx == 3 || s.idx == 5 || s.idx == 7) { bnw[s.idx] = p + o; o += s.cout; bnb[s.idx] = p + o; o += s.cout; }
    }
    auto lstm = [&](std::vector<LstmW<T>>& v, int in0, int H, int layers) {
      v.clear();
      for (int l = 0; l < layers; ++l) {
        LstmW<T> w; w.in = l == 0 ? in0 : H; w.H = H;
        w.wi = p + o; o += (size_t)4 * H * w.in; w.bi = p + o; o += 4 * H; w.wh = p + o; o += (size_t)4 * H * H; w.bh = p + o; o += 4 * H;
        v.push_back(w);
      }
    };
    const int He = c.enc_hidden, Hd = 2 * He;
    lstm(enc[0], 512, He, c.enc_layers); lstm(enc[1], 512, He, c.enc_layers);
    lookup = p + o; o += (size_t)c.vocab * c.emb;
    lstm(dec, c.emb + (c.input_feed ? Hd : 0), Hd, c.dec_layers);
    wa = p + o; o += (size_t)Hd * Hd; wc = p + o; o += (size_t)Hd * 2 * Hd;
    wo = p + o; o += (size_t)c.vocab * Hd; bo = p + o; o += c.vocab;
    total = o;
  }
};
size_t param_count(const Cfg& c) { Params<float> p; p.bind(nullptr, c); return p.total; }
// group boundaries for optim.sgd_list (model.lua:150: cnn, enc_fw, enc_bw, decoder, projector)
void group_offsets(const Cfg& c, size_t off[6]) {
  Params<float> p; p.bind(nullptr, c);
  off[0] = 0; off[1] = p.enc[0][0].wi - (float*)nullptr; off[2] = p.enc[1][0].wi - (float*)nullptr; off[3] = p.lookup - (float*)nullptr;
  off[4] = p.wo - (float*)nullptr; off[5] = p.total;
}

// ------------------------------------------------------------------------------------------------ CNN (cnn.lua:9-45), NCHW
template <class T> struct Map { int C = 0, H = 0, W = 0; std::vector<T> v; size_t img() const { return (size_t)C * H * W; } };
template <class T> struct CnnCache {
  Map<T> in[8];          // input of conv i (post previous activation)
  Map<T> y[8];           // conv i output (pre BN / ReLU)
  Map<T> act[8];         // after (BN), ReLU (pre-pool)
  std::vector<int32_t> pidx[8];   // arg-max of the pool after conv i
  std::vector<T> mean[8], invstd[8];
  int B;
};

template <class T> void im2col(const T* x, int C, int H, int W, int k, int pad, int Ho, int Wo, T* cols) {   // [C*k*k][Ho*Wo]
  for (int c = 0; c < C; ++c) for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) {
    T* row = cols + ((size_t)(c * k + kh) * k + kw) * Ho * Wo;
    for (int oh = 0; oh < Ho; ++oh) {
      const int ih = oh + kh - pad;
      for (int ow = 0; ow < Wo; ++ow) {
        const int iw = ow + kw - pad;
        row[oh * Wo + ow] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((size_t)c * H + ih) * W + iw] : T(0);
      }
    }
  }
}
template <class T> void im2row(const T* x, int C, int H, int W, int k, int pad, int Ho, int Wo, T* rows, bool par) {   // [Ho*Wo][C*k*k]
  const int CK = C * k * k;
#pragma omp parallel for schedule(static) if (par)
  for (int p = 0; p < Ho * Wo; ++p) {
    const int oh = p / Wo, ow = p % Wo; T* r = rows + (size_t)p * CK;
    for (int c = 0; c < C; ++c) for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) {
      const int ih = oh + kh - pad, iw = ow + kw - pad;
      r[(c * k + kh) * k + kw] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((size_t)c * H + ih) * W + iw] : T(0);
    }
  }
}
template <class T> void col2im_add(const T* cols, int C, int H, int W, int k, int pad, int Ho, int Wo, T* dx) {
  for (int c = 0; c < C; ++c) for (int kh = 0; kh < k; ++kh) for (int kw = 0; kw < k; ++kw) {
    const T* row = cols + ((size_t)(c * k + kh) * k + kw) * Ho * Wo;
    for (int oh = 0; oh < Ho; ++oh) {
      const int ih = oh + kh - pad; if (ih < 0 || ih >= H) continue;
      for (int ow = 0; ow < Wo; ++ow) { const int iw = ow + kw - pad; if (iw >= 0 && iw < W) dx[((size_t)c * H + ih) * W + iw] += row[oh * Wo + ow]; }
    }
  }
}

template <class T> void conv_forward(const Map<T>& x, int B, const ConvSpec& s, const T* W, const T* bias, Map<T>& y) {
  const int Ho = x.H + 2 * s.pad - s.k + 1, Wo = x.W + 2 * s.pad - s.k + 1, CK = s.cin * s.k * s.k, HW = Ho * Wo;
  y.C = s.cout; y.H = Ho; y.W = Wo; y.v.resize((size_t)B * s.cout * HW);
  const bool over_images = B >= 4;                         // images are independent: one image per thread, serial GEMM inside
#pragma omp parallel if (over_images) num_threads(std::min(B, nthreads()))
  {
    std::vector<T> cols((size_t)CK * HW);
#pragma omp for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
      T* yb = y.v.data() + (size_t)b * s.cout * HW;
      im2col(x.v.data() + (size_t)b * x.img(), s.cin, x.H, x.W, s.k, s.pad, Ho, Wo, cols.data());
      for (int co = 0; co < s.cout; ++co) std::fill(yb + (size_t)co * HW, yb + (size_t)(co + 1) * HW, bias[co]);     // output = bias, then addmm
      gemm(s.cout, HW, CK, W, CK, 1, cols.data(), HW, yb, HW, true, !over_images);
    }
  }
}
// gradInput (may be null), gradWeight +=, gradBias +=
template <class T> void conv_backward(const Map<T>& x, int B, const ConvSpec& s, const T* W, const std::vector<T>& dy, T* dW, T* db, std::vector<T>* dx) {
  const int Ho = x.H + 2 * s.pad - s.k + 1, Wo = x.W + 2 * s.pad - s.k + 1, CK = s.cin * s.k * s.k, HW = Ho * Wo;
  const bool over_images = B >= 4;
  if (dx) {
    dx->assign((size_t)B * x.img(), T(0));
#pragma omp parallel if (over_images) num_threads(std::min(B, nthreads()))
    {
      std::vector<T> dcols((size_t)CK * HW);
#pragma omp for schedule(dynamic)
      for (int b = 0; b < B; ++b) {
        gemm(CK, HW, s.cout, W, 1, CK, dy.data() + (size_t)b * s.cout * HW, HW, dcols.data(), HW, false, !over_images);   // W^T gradOutput
        col2im_add(dcols.data(), s.cin, x.H, x.W, s.k, s.pad, Ho, Wo, dx->data() + (size_t)b * x.img());
      }
    }
  }
  // accGradParameters, image by image.  With several images each thread sums its images into a private gradWeight and the
  // threads' sums are added up afterwards (the reference's order is image 1..B into one buffer; fp addition order differs only here).
  const int team = over_images ? std::min(std::min(B, nthreads()), 16) : 1;
  std::vector<std::vector<T>> acc(team);
#pragma omp parallel num_threads(team) if (team > 1)
  {
#ifdef _OPENMP
    const int tid = team > 1 ? omp_get_thread_num() : 0;
#else
    const int tid = 0;
#endif
    std::vector<T> rows((size_t)HW * CK);
    std::vector<T>& a = acc[tid]; a.assign((size_t)s.cout * CK + s.cout, T(0));
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      im2row(x.v.data() + (size_t)b * x.img(), s.cin, x.H, x.W, s.k, s.pad, Ho, Wo, rows.data(), team == 1);
      const T* dyb = dy.data() + (size_t)b * s.cout * HW;
      gemm(s.cout, CK, HW, dyb, HW, 1, rows.data(), CK, a.data(), CK, true, team == 1);
      for (int co = 0; co < s.cout; ++co) { T sacc = 0; for (int p = 0; p < HW; ++p) sacc += dyb[(size_t)co * HW + p]; a[(size_t)s.cout * CK + co] += sacc; }
    }
  }
  const size_t nW = (size_t)s.cout * CK;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < nW; ++i) { T v = 0; for (int t = 0; t < team; ++t) v += acc[t][i]; dW[i] += v; }
  for (int co = 0; co < s.cout; ++co) { T v = 0; for (int t = 0; t < team; ++t) v += acc[t][nW + co]; db[co] += v; }
}

template <class T> void relu_inplace(std::vector<T>& v) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < v.size(); ++i) v[i] = v[i] > T(0) ? v[i] : T(0);
}
template <class T> void maxpool_forward(const Map<T>& x, int B, int kh, int kw, Map<T>& y, std::vector<int32_t>& idx) {
  y.C = x.C; y.H = x.H / kh; y.W = x.W / kw;            // floor mode, stride = kernel
  y.v.resize((size_t)B * y.img()); idx.resize(y.v.size());
#pragma omp parallel for schedule(static)
  for (long bc = 0; bc < (long)B * x.C; ++bc) {
    const T* xp = x.v.data() + (size_t)bc * x.H * x.W; T* yp = y.v.data() + (size_t)bc * y.H * y.W; int32_t* ip = idx.data() + (size_t)bc * y.H * y.W;
    for (int oh = 0; oh < y.H; ++oh) for (int ow = 0; ow < y.W; ++ow) {
      int best = (oh * kh) * x.W + ow * kw; T bv = xp[best];
      for (int a = 0; a < kh; ++a) for (int c = 0; c < kw; ++c) { const int p = (oh * kh + a) * x.W + ow * kw + c; if (xp[p] > bv) { bv = xp[p]; best = p; } }
      yp[oh * y.W + ow] = bv; ip[oh * y.W + ow] = best;
    }
  }
}
template <class T> void maxpool_backward(const std::vector<T>& dy, const std::vector<int32_t>& idx, int B, int C, int Hy, int Wy, int Hx, int Wx, std::vector<T>& dx) {
  dx.assign((size_t)B * C * Hx * Wx, T(0));
#pragma omp parallel for schedule(static)
  for (long bc = 0; bc < (long)B * C; ++bc)
    for (int p = 0; p < Hy * Wy; ++p) dx[(size_t)bc * Hx * Wx + idx[(size_t)bc * Hy * Wy + p]] += dy[(size_t)bc * Hy * Wy + p];
}
// nn.SpatialBatchNormalization [upstream THNN/BatchNormalization.c]
template <class T> void bn_forward(Map<T>& x, int B, const T* w, const T* bias, T* rm, T* rv, bool training, std::vector<T>& mean, std::vector<T>& invstd) {
  const int C = x.C, HW = x.H * x.W; const long n = (long)B * HW;
  mean.assign(C, 0); invstd.assign(C, 0);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    T m, is;
    if (training) {
      T s = 0; for (int b = 0; b < B; ++b) { const T* p = x.v.data() + ((size_t)b * C + c) * HW; for (int i = 0; i < HW; ++i) s += p[i]; }
      m = s / n;
      T ss = 0; for (int b = 0; b < B; ++b) { const T* p = x.v.data() + ((size_t)b * C + c) * HW; for (int i = 0; i < HW; ++i) ss += (p[i] - m) * (p[i] - m); }
      is = T(1) / std::sqrt(ss / n + T(1e-5));
      rm[c] = T(0.9) * rm[c] + T(0.1) * m;
      rv[c] = T(0.9) * rv[c] + T(0.1) * (ss / (n - 1));
    } else { m = rm[c]; is = T(1) / std::sqrt(rv[c] + T(1e-5)); }
    mean[c] = m; invstd[c] = is;
    for (int b = 0; b < B; ++b) { T* p = x.v.data() + ((size_t)b * C + c) * HW; for (int i = 0; i < HW; ++i) p[i] = (p[i] - m) * is * w[c] + bias[c]; }
  }
}
// training-mode backward: x = the conv output (pre-BN), dy in/out
template <class T> void bn_backward(const Map<T>& x, int B, const T* w, const std::vector<T>& mean, const std::vector<T>& invstd, std::vector<T>& dy, T* dw, T* db) {
  const int C = x.C, HW = x.H * x.W; const long n = (long)B * HW;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    T sdy = 0, dot = 0;
    for (int b = 0; b < B; ++b) { const size_t o = ((size_t)b * C + c) * HW; for (int i = 0; i < HW; ++i) { sdy += dy[o + i]; dot += dy[o + i] * (x.v[o + i] - mean[c]); } }
    const T k = dot * invstd[c] * invstd[c] / n, gm = sdy / n;
    for (int b = 0; b < B; ++b) { const size_t o = ((size_t)b * C + c) * HW; for (int i = 0; i < HW; ++i) dy[o + i] = (dy[o + i] - gm - (x.v[o + i] - mean[c]) * k) * invstd[c] * w[c]; }
    dw[c] += dot * invstd[c]; db[c] += sdy;
  }
}

template <class T> void cnn_forward(const Params<T>& P, T* bn_state, const T* images, int B, int H, int W, bool training, CnnCache<T>& cc, std::vector<T>& feats, int& Tlen) {
  cc.B = B;
  Map<T> cur; cur.C = 1; cur.H = H; cur.W = W; cur.v.resize((size_t)B * H * W);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < cur.v.size(); ++i) cur.v[i] = (images[i] + T(-128)) * (T(1) / T(128));          // cnn.lua:9-10
  size_t bo = 0;
  for (int li = 0; li < 7; ++li) {
    const ConvSpec& s = CONVS[li]; const int i = s.idx;
    cc.in[i] = cur;
    conv_forward(cc.in[i], B, s, P.cw[i], P.cb[i], cc.y[i]);
    Map<T> a = cc.y[i];
    if (i == 3 || i == 5 || i == 7) { bn_forward(a, B, P.bnw[i], P.bnb[i], bn_state + bo, bn_state + bo + s.cout, training, cc.mean[i], cc.invstd[i]); bo += 2 * s.cout; }
    relu_inplace(a.v);
    cc.act[i] = a;
    if (i == 1 || i == 2) maxpool_forward(cc.act[i], B, 2, 2, cur, cc.pidx[i]);                              // cnn.lua:15,20
    else if (i == 4 || i == 6) maxpool_forward(cc.act[i], B, 2, 1, cur, cc.pidx[i]);                         // cnn.lua:29,38 (kW=1,kH=2)
    else cur = cc.act[i];
  }
  Tlen = cur.H * cur.W;                                    // View(512,-1) + Transpose(2,3), cnn.lua:44-45
  feats.resize((size_t)B * Tlen * 512);
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) for (int c = 0; c < 512; ++c) for (int t = 0; t < Tlen; ++t) feats[((size_t)b * Tlen + t) * 512 + c] = cur.v[((size_t)b * 512 + c) * Tlen + t];
}
template <class T> void cnn_backward(const Params<T>& P, Params<T>& G, const CnnCache<T>& cc, const std::vector<T>& dfeats, int Tlen) {
  const int B = cc.B;
  std::vector<T> d((size_t)B * 512 * Tlen), tmp;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < B; ++b) for (int c = 0; c < 512; ++c) for (int t = 0; t < Tlen; ++t) d[((size_t)b * 512 + c) * Tlen + t] = dfeats[((size_t)b * Tlen + t) * 512 + c];
  for (int li = 6; li >= 0; --li) {
    const ConvSpec& s = CONVS[li]; const int i = s.idx;
    const Map<T>& a = cc.act[i];
    if (i == 1 || i == 2 || i == 4 || i == 6) {
      const int kh = 2, kw = (i <= 2) ? 2 : 1;
      maxpool_backward(d, cc.pidx[i], B, a.C, a.H / kh, a.W / kw, a.H, a.W, tmp); d.swap(tmp);
    }
#pragma omp parallel for schedule(static)
    for (size_t j = 0; j < d.size(); ++j) if (!(a.v[j] > T(0))) d[j] = T(0);                                  // ReLU: gradInput = gradOutput where output > 0
    if (i == 3 || i == 5 || i == 7) bn_backward(cc.y[i], B, P.bnw[i], cc.mean[i], cc.invstd[i], d, G.bnw[i], G.bnb[i]);
    conv_backward(cc.in[i], B, s, P.cw[i], d, G.cw[i], G.cb[i], li > 0 ? &tmp : nullptr);   // model.lua:692 computes d(input) of conv1 too; it is unused, skipped here
    if (li > 0) d.swap(tmp);
  }
}

// ------------------------------------------------------------------------------------------------ LSTM cell (LSTM.lua:79-105)
template <class T> inline T sigm(T x) { return T(1) / (T(1) + std::exp(-x)); }
template <class T> struct CellCache { std::vector<T> x, c_prev, h_prev, gates /*[B][4H] i f o g*/, c, h; };

template <class T> void cell_forward(const LstmW<T>& w, const Linear<T>& li, const Linear<T>& lh, const T* x, long ldx, const T* c_prev, const T* h_prev, int B, CellCache<T>& cc) {
  const int H = w.H;
  std::vector<T> z((size_t)B * 4 * H);
  for (int b = 0; b < B; ++b) for (int n = 0; n < 4 * H; ++n) z[(size_t)b * 4 * H + n] = w.bi[n];             // Linear: output = bias, then addmm
  li.fwd(x, ldx, B, z.data(), 4 * H, true);
  std::vector<T> z2((size_t)B * 4 * H);
  for (int b = 0; b < B; ++b) for (int n = 0; n < 4 * H; ++n) z2[(size_t)b * 4 * H + n] = w.bh[n];
  lh.fwd(h_prev, H, B, z2.data(), 4 * H, true);
  cc.x.resize((size_t)B * w.in); for (int b = 0; b < B; ++b) std::copy(x + (size_t)b * ldx, x + (size_t)b * ldx + w.in, cc.x.begin() + (size_t)b * w.in);
  cc.c_prev.assign(c_prev, c_prev + (size_t)B * H); cc.h_prev.assign(h_prev, h_prev + (size_t)B * H);
  cc.gates.resize((size_t)B * 4 * H); cc.c.resize((size_t)B * H); cc.h.resize((size_t)B * H);
#pragma omp parallel for schedule(static) num_threads(small_team(B))
  for (int b = 0; b < B; ++b) for (int j = 0; j < H; ++j) {
    const size_t o = (size_t)b * 4 * H;
    const T ig = sigm(z[o + j] + z2[o + j]), fg = sigm(z[o + H + j] + z2[o + H + j]), og = sigm(z[o + 2 * H + j] + z2[o + 2 * H + j]);
    const T gg = std::tanh(z[o + 3 * H + j] + z2[o + 3 * H + j]);                                            // CAddTable, Reshape(4,H), SplitTable: n1..n4
    cc.gates[o + j] = ig; cc.gates[o + H + j] = fg; cc.gates[o + 2 * H + j] = og; cc.gates[o + 3 * H + j] = gg;
    const T c = fg * c_prev[(size_t)b * H + j] + ig * gg;
    cc.c[(size_t)b * H + j] = c; cc.h[(size_t)b * H + j] = og * std::tanh(c);
  }
}
// dc/dh in: gradient wrt (c_out, h_out); out: dz (B,4H), dc_prev in place of dc
template <class T> void cell_backward_gates(const CellCache<T>& cc, int B, int H, std::vector<T>& dc, const std::vector<T>& dh, std::vector<T>& dz) {
  dz.resize((size_t)B * 4 * H);
#pragma omp parallel for schedule(static) num_threads(small_team(B))
  for (int b = 0; b < B; ++b) for (int j = 0; j < H; ++j) {
    const size_t o = (size_t)b * 4 * H, s = (size_t)b * H + j;
    const T ig = cc.gates[o + j], fg = cc.gates[o + H + j], og = cc.gates[o + 2 * H + j], gg = cc.gates[o + 3 * H + j];
    const T tc = std::tanh(cc.c[s]);
    const T dct = dc[s] + dh[s] * og * (T(1) - tc * tc);
    dz[o + j] = dct * gg * ig * (T(1) - ig);
    dz[o + H + j] = dct * cc.c_prev[s] * fg * (T(1) - fg);
    dz[o + 2 * H + j] = dh[s] * tc * og * (T(1) - og);
    dz[o + 3 * H + j] = dct * ig * (T(1) - gg * gg);
    dc[s] = dct * fg;
  }
}

// ------------------------------------------------------------------------------------------------ attention (LSTM.lua:124-162)
template <class T> struct AttnCache { std::vector<T> q, a, cat, out; };
template <class T> void attn_forward(const Linear<T>& la, const Linear<T>& lc, const T* h_top, const T* ctx, int R, int ctx_div, int Tn, int Hd, AttnCache<T>& ac) {
  ac.q.resize((size_t)R * Hd); ac.a.resize((size_t)R * Tn); ac.cat.resize((size_t)R * 2 * Hd); ac.out.resize((size_t)R * Hd);
  la.fwd(h_top, Hd, R, ac.q.data(), Hd, false);                                                              // LinearNoBias, :131
#pragma omp parallel for schedule(static) num_threads(small_team(R))
  for (int r = 0; r < R; ++r) {
    const T* cx = ctx + (size_t)(r / ctx_div) * Tn * Hd; T* a = ac.a.data() + (size_t)r * Tn; const T* q = ac.q.data() + (size_t)r * Hd;
    T mx = -INFINITY;
    for (int t = 0; t < Tn; ++t) { T s = 0; for (int j = 0; j < Hd; ++j) s += cx[(size_t)t * Hd + j] * q[j]; a[t] = s; mx = std::max(mx, s); }   // MM + Sum, :135-138
    T den = 0; for (int t = 0; t < Tn; ++t) { a[t] = std::exp(a[t] - mx); den += a[t]; }
    for (int t = 0; t < Tn; ++t) a[t] /= den;                                                                // SoftMax, :139
    T* cat = ac.cat.data() + (size_t)r * 2 * Hd;
    for (int j = 0; j < Hd; ++j) cat[j] = 0;
    for (int t = 0; t < Tn; ++t) for (int j = 0; j < Hd; ++j) cat[j] += a[t] * cx[(size_t)t * Hd + j];       // MM, :145-150
    for (int j = 0; j < Hd; ++j) cat[Hd + j] = h_top[(size_t)r * Hd + j];                                    // JoinTable, :153
  }
  lc.fwd(ac.cat.data(), 2 * Hd, R, ac.out.data(), Hd, false);                                                // LinearNoBias, :155
  for (auto& v : ac.out) v = std::tanh(v);
}

template <class T> void logsoftmax_rows(const T* logits, int R, int V, T* logp) {
  for (int r = 0; r < R; ++r) {
    T mx = -INFINITY; for (int v = 0; v < V; ++v) mx = std::max(mx, logits[(size_t)r * V + v]);
    T s = 0; for (int v = 0; v < V; ++v) s += std::exp(logits[(size_t)r * V + v] - mx);
    const T lse = mx + std::log(s);
    for (int v = 0; v < V; ++v) logp[(size_t)r * V + v] = logits[(size_t)r * V + v] - lse;
  }
}

// ------------------------------------------------------------------------------------------------ the model
template <class T> struct Model {
  Cfg cfg; int He, Hd, Le, Ld, E, V;
  Params<T> P;
  std::vector<Linear<T>> enc_li[2], enc_lh[2], dec_li, dec_lh; Linear<T> la, lc, lo;
  void prepare(const Cfg& c, T* params) {
    cfg = c; He = c.enc_hidden; Hd = 2 * He; Le = c.enc_layers; Ld = c.dec_layers; E = c.emb; V = c.vocab;
    P.bind(params, c);
    for (int d = 0; d < 2; ++d) { enc_li[d].resize(Le); enc_lh[d].resize(Le); for (int l = 0; l < Le; ++l) { enc_li[d][l].prepare(P.enc[d][l].wi, 4 * He, P.enc[d][l].in); enc_lh[d][l].prepare(P.enc[d][l].wh, 4 * He, He); } }
    dec_li.resize(Ld); dec_lh.resize(Ld);
    for (int l = 0; l < Ld; ++l) { dec_li[l].prepare(P.dec[l].wi, 4 * Hd, P.dec[l].in); dec_lh[l].prepare(P.dec[l].wh, 4 * Hd, Hd); }
    la.prepare(P.wa, Hd, Hd); lc.prepare(P.wc, Hd, 2 * Hd); lo.prepare(P.wo, V, Hd);
  }

  // encoder, model.lua:291-316.  traces[dir][t][layer]
  std::vector<std::vector<std::vector<CellCache<T>>>> etr;
  std::vector<T> context;
  void encoder_forward(const std::vector<T>& feats, int B, int Tn) {
    etr.assign(2, {}); context.assign((size_t)B * Tn * Hd, 0);
    for (int d = 0; d < 2; ++d) {
      etr[d].assign(Tn, std::vector<CellCache<T>>(Le));
      std::vector<std::vector<T>> c(Le, std::vector<T>((size_t)B * He, 0)), h(Le, std::vector<T>((size_t)B * He, 0));   // reset_state(...,0), :293,305
      for (int it = 0; it < Tn; ++it) {
        const int t = d == 0 ? it : Tn - 1 - it;
        const T* x = feats.data() + (size_t)t * 512; long ldx = (long)Tn * 512;
        for (int l = 0; l < Le; ++l) {
          CellCache<T>& cc = etr[d][t][l];
          cell_forward(P.enc[d][l], enc_li[d][l], enc_lh[d][l], x, ldx, c[l].data(), h[l].data(), B, cc);
          c[l] = cc.c; h[l] = cc.h; x = cc.h.data(); ldx = He;                                                 // Dropout(0) between layers = identity (S6)
        }
        for (int b = 0; b < B; ++b) std::copy(h[Le - 1].begin() + (size_t)b * He, h[Le - 1].begin() + (size_t)(b + 1) * He,
                                              context.begin() + ((size_t)b * Tn + t) * Hd + d * He);           // :303,315
      }
    }
  }
  // decoder initial state, model.lua:539-552 (+ quirk S5)
  void dec_init(int B, int Tn, std::vector<std::vector<T>>& c, std::vector<std::vector<T>>& h) {
    c.assign(Ld, std::vector<T>((size_t)B * Hd, 0)); h.assign(Ld, std::vector<T>((size_t)B * Hd, 0));
    const CellCache<T>& ff = etr[0][Tn - 1][Le - 1]; const CellCache<T>& bb = etr[1][0][Le - 1];
    for (int b = 0; b < B; ++b) for (int j = 0; j < He; ++j) {
      c[0][(size_t)b * Hd + j] = ff.c[(size_t)b * He + j]; c[0][(size_t)b * Hd + He + j] = bb.c[(size_t)b * He + j];
      h[0][(size_t)b * Hd + j] = ff.h[(size_t)b * He + j]; h[0][(size_t)b * Hd + He + j] = bb.h[(size_t)b * He + j];
    }
    if (cfg.input_feed && Ld >= 2) std::fill(h[0].begin(), h[0].end(), T(0));       // the loop of :549-552 uses offset +0: it zeroes h1(0) (and c2(0))
  }
  struct DecStep { std::vector<CellCache<T>> cells; AttnCache<T> at; std::vector<T> xin; };
  // one decoder clone forward (LSTM.lua:18-122): tok (R) 1-based, feed (R,Hd), state c/h per layer (updated in place)
  void dec_step(const int32_t* tok, long tok_stride, const std::vector<T>& feed, std::vector<std::vector<T>>& c, std::vector<std::vector<T>>& h, int R, int ctx_div, int Tn, DecStep& st) {
    const int in0 = P.dec[0].in;
    st.xin.resize((size_t)R * in0);
    for (int r = 0; r < R; ++r) {
      const T* e = P.lookup + (size_t)(tok[(size_t)r * tok_stride] - 1) * E;                                  // LookupTable, LSTM.lua:55
      std::copy(e, e + E, st.xin.begin() + (size_t)r * in0);
      if (cfg.input_feed) std::copy(feed.begin() + (size_t)r * Hd, feed.begin() + (size_t)(r + 1) * Hd, st.xin.begin() + (size_t)r * in0 + E);   // JoinTable, :59-64
    }
    st.cells.resize(Ld);
    const T* x = st.xin.data(); long ldx = in0;
    for (int l = 0; l < Ld; ++l) {
      cell_forward(P.dec[l], dec_li[l], dec_lh[l], x, ldx, c[l].data(), h[l].data(), R, st.cells[l]);
      c[l] = st.cells[l].c; h[l] = st.cells[l].h; x = st.cells[l].h.data(); ldx = Hd;
    }
    attn_forward(la, lc, h[Ld - 1].data(), context.data(), R, ctx_div, Tn, Hd, st.at);
  }
  void project(const std::vector<T>& out, int R, std::vector<T>& logits, std::vector<T>& logp) {
    logits.resize((size_t)R * V); logp.resize((size_t)R * V);
    for (int r = 0; r < R; ++r) for (int v = 0; v < V; ++v) logits[(size_t)r * V + v] = P.bo[v];
    lo.fwd(out.data(), Hd, R, logits.data(), V, true);
    logsoftmax_rows(logits.data(), R, V, logp.data());
  }
};

struct PhaseTimer {                      // AOCR_CPU_REF_TIMING=1 prints where a step spends its time
  bool on = getenv("AOCR_CPU_REF_TIMING") != nullptr; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  void lap(const char* what) { if (!on) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[cpu_ref] %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; }
};

template <class T>
int train_step(const Cfg& cfg, T* params, T* bn_state, const T* images, const int32_t* tgt, const int32_t* tge, int B, int W, int L,
               T* loss_out, T* logits_out, T* grads, T* context_out, T* feats_out) {
  Model<T> M; M.prepare(cfg, params);
  const int Hd = M.Hd, He = M.He, Ld = M.Ld, Le = M.Le, V = M.V, E = M.E;
  CnnCache<T> cc; std::vector<T> feats; int Tn = 0;
  PhaseTimer pt;
  cnn_forward(M.P, bn_state, images, B, cfg.img_h, W, true, cc, feats, Tn);                                  // model.lua:285
  pt.lap("cnn forward");
  M.encoder_forward(feats, B, Tn);
  pt.lap("encoder forward");
  if (feats_out) std::copy(feats.begin(), feats.end(), feats_out);
  if (context_out) std::copy(M.context.begin(), M.context.end(), context_out);
  std::vector<std::vector<T>> c, h; M.dec_init(B, Tn, c, h);
  std::vector<T> feed((size_t)B * Hd, 0);
  std::vector<typename Model<T>::DecStep> steps(L);
  for (int t = 0; t < L; ++t) {                                                                              // model.lua:553-568
    M.dec_step(tgt + t, L, feed, c, h, B, 1, Tn, steps[t]);
    if (cfg.input_feed) feed = steps[t].at.out;
  }
  pt.lap("decoder forward");
  // ---- backward, model.lua:634-694
  Params<T> G; std::fill(grads, grads + M.P.total, T(0)); G.bind(grads, cfg);
  std::vector<T> dctx((size_t)B * Tn * Hd, 0);
  std::vector<std::vector<T>> dc(Ld, std::vector<T>((size_t)B * Hd, 0)), dh(Ld, std::vector<T>((size_t)B * Hd, 0));
  std::vector<T> dfeed((size_t)B * Hd, 0), logits, logp, dlogits((size_t)B * V), dout((size_t)B * Hd), dpre((size_t)B * Hd), dcat((size_t)B * 2 * Hd), dq((size_t)B * Hd), dz, dx;
  T loss = 0;
  for (int t = L - 1; t >= 0; --t) {
    typename Model<T>::DecStep& st = steps[t];
    M.project(st.at.out, B, logits, logp);                                                                   // :644 (projector forward again)
    if (logits_out) std::copy(logits.begin(), logits.end(), logits_out + (size_t)t * B * V);
    T nll = 0;
    for (int b = 0; b < B; ++b) {
      const int y = tge[(size_t)b * L + t] - 1; const T wy = (y + 1 == PAD) ? T(0) : T(1);                   // criterion.lua:4-5
      nll -= wy * logp[(size_t)b * V + y];
      // d logp = -w[y]/B at y (:646-647); LogSoftMax backward: dlogits = dlogp - exp(logp) * sum(dlogp)
      const T g = -wy / B;
      for (int v = 0; v < V; ++v) dlogits[(size_t)b * V + v] = (v == y ? g : T(0)) - std::exp(logp[(size_t)b * V + v]) * g;
    }
    loss += nll / B;                                                                                         // :645
    linear_bwd_weight(dlogits.data(), V, st.at.out.data(), Hd, B, V, Hd, G.wo); colsum_add(dlogits.data(), V, B, V, G.bo);
    linear_bwd_input(dlogits.data(), V, M.P.wo, B, V, Hd, dout.data(), Hd, false);
    for (size_t i = 0; i < dout.size(); ++i) dout[i] += dfeed[i];                                            // :649
    // attention block backward (nngraph reverse order of LSTM.lua:131-157)
    for (size_t i = 0; i < dpre.size(); ++i) dpre[i] = dout[i] * (T(1) - st.at.out[i] * st.at.out[i]);
    linear_bwd_weight(dpre.data(), Hd, st.at.cat.data(), 2 * Hd, B, Hd, 2 * Hd, G.wc);
    linear_bwd_input(dpre.data(), Hd, M.P.wc, B, Hd, 2 * Hd, dcat.data(), 2 * Hd, false);
#pragma omp parallel for schedule(static) num_threads(small_team(B))
    for (int b = 0; b < B; ++b) {
      const T* cx = M.context.data() + (size_t)b * Tn * Hd; T* dcx = dctx.data() + (size_t)b * Tn * Hd;
      const T* a = st.at.a.data() + (size_t)b * Tn; const T* dcv = dcat.data() + (size_t)b * 2 * Hd; const T* q = st.at.q.data() + (size_t)b * Hd;
      std::vector<T> da(Tn), ds(Tn);
      T dot = 0;
      for (int tt = 0; tt < Tn; ++tt) { T s = 0; for (int j = 0; j < Hd; ++j) s += cx[(size_t)tt * Hd + j] * dcv[j]; da[tt] = s; dot += a[tt] * s; }
      for (int tt = 0; tt < Tn; ++tt) ds[tt] = a[tt] * (da[tt] - dot);                                       // SoftMax backward
      T* dqb = dq.data() + (size_t)b * Hd; for (int j = 0; j < Hd; ++j) dqb[j] = 0;
      for (int tt = 0; tt < Tn; ++tt) for (int j = 0; j < Hd; ++j) {
        dcx[(size_t)tt * Hd + j] += a[tt] * dcv[j] + ds[tt] * q[j];                                          // model.lua:652-653 accumulates d(context)
        dqb[j] += ds[tt] * cx[(size_t)tt * Hd + j];
      }
    }
    linear_bwd_weight(dq.data(), Hd, st.cells[Ld - 1].h.data(), Hd, B, Hd, Hd, G.wa);
    std::vector<T> dtop((size_t)B * Hd);
    linear_bwd_input(dq.data(), Hd, M.P.wa, B, Hd, Hd, dtop.data(), Hd, false);
    for (int b = 0; b < B; ++b) for (int j = 0; j < Hd; ++j) dh[Ld - 1][(size_t)b * Hd + j] += dtop[(size_t)b * Hd + j] + dcat[(size_t)b * 2 * Hd + Hd + j];
    for (int l = Ld - 1; l >= 0; --l) {
      const LstmW<T>& w = M.P.dec[l]; const CellCache<T>& cl = st.cells[l];
      cell_backward_gates(cl, B, Hd, dc[l], dh[l], dz);
      linear_bwd_weight(dz.data(), 4 * Hd, cl.x.data(), w.in, B, 4 * Hd, w.in, G.dec[l].wi); colsum_add(dz.data(), 4 * Hd, B, 4 * Hd, G.dec[l].bi);
      linear_bwd_weight(dz.data(), 4 * Hd, cl.h_prev.data(), Hd, B, 4 * Hd, Hd, G.dec[l].wh); colsum_add(dz.data(), 4 * Hd, B, 4 * Hd, G.dec[l].bh);
      dx.resize((size_t)B * w.in);
      linear_bwd_input(dz.data(), 4 * Hd, w.wi, B, 4 * Hd, w.in, dx.data(), w.in, false);
      linear_bwd_input(dz.data(), 4 * Hd, w.wh, B, 4 * Hd, Hd, dh[l].data(), Hd, false);                     // d h_prev replaces d h
      if (l > 0) for (size_t i = 0; i < dx.size(); ++i) dh[l - 1][i] += dx[i];
    }
    for (int b = 0; b < B; ++b) {                                                                            // LookupTable accGradParameters
      T* dl = G.lookup + (size_t)(tgt[(size_t)b * L + t] - 1) * E;
      for (int j = 0; j < E; ++j) dl[j] += dx[(size_t)b * M.P.dec[0].in + j];
      if (cfg.input_feed) for (int j = 0; j < Hd; ++j) dfeed[(size_t)b * Hd + j] = dx[(size_t)b * M.P.dec[0].in + E + j];   // :654-657
    }
    if (!cfg.input_feed) std::fill(dfeed.begin(), dfeed.end(), T(0));
  }
  pt.lap("decoder backward");
  // encoder BPTT, model.lua:662-690
  std::vector<T> dfeats((size_t)B * Tn * 512, 0);
  for (int d = 0; d < 2; ++d) {
    std::vector<std::vector<T>> dce(Le, std::vector<T>((size_t)B * He, 0)), dhe(Le, std::vector<T>((size_t)B * He, 0));
    for (int b = 0; b < B; ++b) for (int j = 0; j < He; ++j) {
      dce[Le - 1][(size_t)b * He + j] = dc[0][(size_t)b * Hd + d * He + j];                                  // :666 / :680
      dhe[Le - 1][(size_t)b * He + j] = dh[0][(size_t)b * Hd + d * He + j];                                  // :667 / :681 (quirk S5: passed on although h1(0) was zeroed)
    }
    for (int it = 0; it < Tn; ++it) {
      const int t = d == 0 ? Tn - 1 - it : it;
      for (int b = 0; b < B; ++b) for (int j = 0; j < He; ++j) dhe[Le - 1][(size_t)b * He + j] += dctx[((size_t)b * Tn + t) * Hd + d * He + j];   // :670 / :684
      for (int l = Le - 1; l >= 0; --l) {
        const LstmW<T>& w = M.P.enc[d][l]; const CellCache<T>& cl = M.etr[d][t][l];
        cell_backward_gates(cl, B, He, dce[l], dhe[l], dz);
        linear_bwd_weight(dz.data(), 4 * He, cl.x.data(), w.in, B, 4 * He, w.in, G.enc[d][l].wi); colsum_add(dz.data(), 4 * He, B, 4 * He, G.enc[d][l].bi);
        linear_bwd_weight(dz.data(), 4 * He, cl.h_prev.data(), He, B, 4 * He, He, G.enc[d][l].wh); colsum_add(dz.data(), 4 * He, B, 4 * He, G.enc[d][l].bh);
        dx.resize((size_t)B * w.in);
        linear_bwd_input(dz.data(), 4 * He, w.wi, B, 4 * He, w.in, dx.data(), w.in, false);
        linear_bwd_input(dz.data(), 4 * He, w.wh, B, 4 * He, He, dhe[l].data(), He, false);
        if (l > 0) for (size_t i = 0; i < dx.size(); ++i) dhe[l - 1][i] += dx[i];
      }
      for (int b = 0; b < B; ++b) for (int j = 0; j < 512; ++j) dfeats[((size_t)b * Tn + t) * 512 + j] += dx[(size_t)b * 512 + j];   // :675 copy / :689 add
    }
  }
  pt.lap("encoder backward");
  cnn_backward(M.P, G, cc, dfeats, Tn);                                                                       // :692
  pt.lap("cnn backward");
  *loss_out = loss * B;                                                                                       // model.lua:701 returns loss * batch_size
  return 0;
}

// optim.sgd_list, optim_sgd.lua:38-95 with wd = mom = 0: per group clip to `clip`, then x -= lr * g.  norms: {param, grad} x 5
template <class T> int sgd(const Cfg& cfg, T* params, T* grads, T lr, T clip, T* norms) {
  size_t off[6]; group_offsets(cfg, off);
  for (int g = 0; g < 5; ++g) {
    T pn = 0, gn = 0;
    for (size_t i = off[g]; i < off[g + 1]; ++i) { pn += params[i] * params[i]; gn += grads[i] * grads[i]; }
    pn = std::sqrt(pn); gn = std::sqrt(gn);
    if (norms) { norms[2 * g] = pn; norms[2 * g + 1] = gn; }
    const T shrink = gn > clip ? clip / gn : T(1);                                                            // :50-52
    for (size_t i = off[g]; i < off[g + 1]; ++i) { grads[i] *= shrink; params[i] -= lr * grads[i]; }          // :90
  }
  return 0;
}

// forward_only step, model.lua:321-627: eval-mode CNN, encoder, beam search (S9 fixed, S10 sorted), back-trace, gold pass.
template <class T>
int decode(const Cfg& cfg, T* params, T* bn_state, const T* images, const int32_t* tgt, const int32_t* tge, int B, int W, int L, int beam, int Lt,
           int32_t* labels, T* scores, T* gold, T* loss_out) {
  Model<T> M; M.prepare(cfg, params);
  const int Hd = M.Hd, Ld = M.Ld, V = M.V; const int k = std::min(beam, V);
  CnnCache<T> cc; std::vector<T> feats; int Tn = 0;
  std::vector<T> bn(bn_state, bn_state + 2 * (256 + 512 + 512));
  cnn_forward(M.P, bn.data(), images, B, cfg.img_h, W, false, cc, feats, Tn);
  M.encoder_forward(feats, B, Tn);
  std::vector<int32_t> tp((size_t)B * Lt, PAD), ep((size_t)B * Lt, PAD);                                     // pad targets to max_decoder_l, :266-274
  for (int b = 0; b < B; ++b) for (int t = 0; t < L && t < Lt; ++t) { tp[(size_t)b * Lt + t] = tgt[(size_t)b * L + t]; ep[(size_t)b * Lt + t] = tge[(size_t)b * L + t]; }
  std::vector<std::vector<T>> c0, h0; M.dec_init(B, Tn, c0, h0);
  std::vector<std::vector<T>> c = c0, h = h0;
  std::vector<T> feed((size_t)B * Hd, 0), bs((size_t)B * k, 0), logits, logp;
  std::vector<int32_t> tok(B); for (int b = 0; b < B; ++b) tok[b] = tp[(size_t)b * Lt];
  std::vector<int32_t> htok((size_t)Lt * B * k), hpar((size_t)Lt * B * k);
  typename Model<T>::DecStep st;
  for (int t = 0; t < Lt; ++t) {
    const int kin = t == 0 ? 1 : k, R = B * kin;
    M.dec_step(tok.data(), 1, feed, c, h, R, kin, Tn, st);
    M.project(st.at.out, R, logits, logp);
    std::vector<int32_t> ntok((size_t)B * k), par((size_t)B * k);
    for (int b = 0; b < B; ++b) {
      std::vector<std::pair<T, int>> cand((size_t)kin * V);
      for (int j = 0; j < kin; ++j) {
        const int r = b * kin + j; const bool fin = t > 0 && (tok[r] == PAD || tok[r] == EOS);
        for (int v = 0; v < V; ++v) {
          T lp = logp[(size_t)r * V + v];
          if (fin && v == PAD - 1) lp = 0;                                                                    // :448-449
          cand[(size_t)j * V + v] = {(t > 0 ? bs[(size_t)b * k + j] : T(0)) + lp, j * V + v};                 // :450
        }
      }
      std::stable_sort(cand.begin(), cand.end(), [](const std::pair<T, int>& x, const std::pair<T, int>& y) { return x.first > y.first; });   // topk, :402,452 (S10)
      for (int j = 0; j < k; ++j) {
        const int raw = cand[j].second;
        bs[(size_t)b * k + j] = cand[j].first; ntok[(size_t)b * k + j] = raw % V + 1; par[(size_t)b * k + j] = t == 0 ? 0 : raw / V;   // :454-458,516 (S9 fixed)
      }
    }
    // gather states by parent, :521-535
    std::vector<std::vector<T>> c2(Ld, std::vector<T>((size_t)B * k * Hd)), h2 = c2; std::vector<T> f2((size_t)B * k * Hd, 0);
    for (int b = 0; b < B; ++b) for (int j = 0; j < k; ++j) {
      const int src = b * kin + par[(size_t)b * k + j], dst = b * k + j;
      for (int l = 0; l < Ld; ++l) {
        std::copy(c[l].begin() + (size_t)src * Hd, c[l].begin() + (size_t)(src + 1) * Hd, c2[l].begin() + (size_t)dst * Hd);
        std::copy(h[l].begin() + (size_t)src * Hd, h[l].begin() + (size_t)(src + 1) * Hd, h2[l].begin() + (size_t)dst * Hd);
      }
      if (cfg.input_feed) std::copy(st.at.out.begin() + (size_t)src * Hd, st.at.out.begin() + (size_t)(src + 1) * Hd, f2.begin() + (size_t)dst * Hd);
    }
    c.swap(c2); h.swap(h2); feed.swap(f2); tok = ntok;
    std::copy(ntok.begin(), ntok.end(), htok.begin() + (size_t)t * B * k); std::copy(par.begin(), par.end(), hpar.begin() + (size_t)t * B * k);
  }
  for (int b = 0; b < B; ++b) {                                                                               // back-trace, :573-585
    int best = 0; for (int j = 1; j < k; ++j) if (bs[(size_t)b * k + j] > bs[(size_t)b * k + best]) best = j;
    scores[b] = bs[(size_t)b * k + best];
    int idx = best;
    for (int t = Lt - 1; t >= 0; --t) { labels[(size_t)b * Lt + t] = htok[((size_t)t * B + b) * k + idx]; idx = hpar[((size_t)t * B + b) * k + idx]; }
  }
  // gold pass, :589-627
  c = c0; h = h0; feed.assign((size_t)B * Hd, 0);
  T loss = 0; for (int b = 0; b < B; ++b) gold[b] = 0;
  for (int t = 0; t < Lt; ++t) {
    M.dec_step(tp.data() + t, Lt, feed, c, h, B, 1, Tn, st);
    M.project(st.at.out, B, logits, logp);
    for (int b = 0; b < B; ++b) {
      const int y = ep[(size_t)b * Lt + t] - 1; const T lp = logp[(size_t)b * V + y];
      if (y + 1 != PAD) { loss -= lp / B; gold[b] += lp; }                                                     // :612, :614-618
    }
    if (cfg.input_feed) feed = st.at.out;
  }
  *loss_out = loss * B;
  return 0;
}

}  // namespace

extern "C" {
typedef Cfg aocr_cpu_ref_cfg;
int64_t aocr_cpu_ref_param_count(const aocr_cpu_ref_cfg* c) { return (int64_t)param_count(*c); }
int aocr_cpu_ref_threads(void) { return nthreads(); }
int aocr_cpu_ref_train_step_f64(const aocr_cpu_ref_cfg* c, double* params, double* bn_state, const double* images, const int32_t* tgt, const int32_t* tge,
                                int32_t B, int32_t W, int32_t L, double* loss, double* logits, double* grads, double* context, double* feats) {
  return train_step<double>(*c, params, bn_state, images, tgt, tge, B, W, L, loss, logits, grads, context, feats);
}
int aocr_cpu_ref_train_step_f32(const aocr_cpu_ref_cfg* c, float* params, float* bn_state, const float* images, const int32_t* tgt, const int32_t* tge,
                                int32_t B, int32_t W, int32_t L, float* loss, float* logits, float* grads, float* context, float* feats) {
  return train_step<float>(*c, params, bn_state, images, tgt, tge, B, W, L, loss, logits, grads, context, feats);
}
int aocr_cpu_ref_sgd_f64(const aocr_cpu_ref_cfg* c, double* params, double* grads, double lr, double clip, double* norms) { return sgd<double>(*c, params, grads, lr, clip, norms); }
int aocr_cpu_ref_sgd_f32(const aocr_cpu_ref_cfg* c, float* params, float* grads, float lr, float clip, float* norms) { return sgd<float>(*c, params, grads, lr, clip, norms); }
int aocr_cpu_ref_decode_f64(const aocr_cpu_ref_cfg* c, double* params, double* bn_state, const double* images, const int32_t* tgt, const int32_t* tge, int32_t B,
                            int32_t W, int32_t L, int32_t beam, int32_t max_decoder_l, int32_t* labels, double* scores, double* gold, double* loss) {
  return decode<double>(*c, params, bn_state, images, tgt, tge, B, W, L, beam, max_decoder_l, labels, scores, gold, loss);
}
int aocr_cpu_ref_decode_f32(const aocr_cpu_ref_cfg* c, float* params, float* bn_state, const float* images, const int32_t* tgt, const int32_t* tge, int32_t B,
                            int32_t W, int32_t L, int32_t beam, int32_t max_decoder_l, int32_t* labels, float* scores, float* gold, float* loss) {
  return decode<float>(*c, params, bn_state, images, tgt, tge, B, W, L, beam, max_decoder_l, labels, scores, gold, loss);
}
}
