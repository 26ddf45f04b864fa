// epilogues.h -- accumulator consumers for mfma_gemm.h kernels.
// quad<NT>(m, n, nstep, v): v[ni][i] is the result at row m+i (i = 0..3, m % 4 == 0) and
// column n + nstep*ni -- or, for the gate epilogues, gate ni of hidden unit n.
#pragma once
#include "mfma_gemm.h"

namespace aocr {

enum : int { EP_RELU = 1, EP_TANH = 2, EP_ACCUM = 4, EP_ATOMIC = 8 };

// Gate activations on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, ~1-2 ulp): the library expf/tanhf cost
// hundreds of VALU instructions per hidden unit and made the gate epilogue -- not the MFMAs -- the longest part of a
// recurrent step.  Absolute error ~1e-7, two orders below the 1e-4 logit tolerance.
__device__ __forceinline__ float fast_exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + fast_exp_(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
  const float ax = fminf(fabsf(x), 15.f);                     // tanh saturates to 1 - 2e-13 by |x| = 15; avoids inf/inf
  const float e = fast_exp_(2.f * ax);
  const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
  return copysignf(t, x);
}

// nn.Dropout (LSTM.lua:68-69 on the input of every layer above the first, :116-118 on the attention output): counter-based mask, the
// same function as oracle_torch.dropout_mask -- keep(idx) = (splitmix64(base + idx) >> 11) >= thr with base = splitmix64(seed ^
// stream * 0xD1342543DE82EF95), stream = 64 * train step + site, thr = ceil(p * 2^53); kept values are scaled by 1 / (1 - p).
// idx = the element's flat offset in its [time][batch][hidden] buffer.  thr == 0: no dropout.  (Kept OUT of EpStore: that struct travels by
// value in the grouped launches' argument arrays, and 40 more bytes made the compiler copy the selected problem to scratch -- the
// grouped weight-gradient kernel went from 128 to 732 us.)
__host__ __device__ __forceinline__ unsigned long long splitmix64_(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct DropSpec {
  unsigned long long base = 0, thr = 0; float scale = 1.f; long long off = 0;       // off: flat offset of row 0 / column 0 of this launch
  __device__ __forceinline__ bool on() const { return thr != 0; }
  __device__ __forceinline__ float mask(long long idx) const { return ((splitmix64_(base + (unsigned long long)(off + idx)) >> 11) >= thr) ? scale : 0.f; }
};

// C[m][n] = act(v + bias[n] + bias2[n]); columns >= N0 go to a second destination (C1, column n-N0).
struct EpStore {
  float* C; int64_t ldc; int M, N;
  const float* bias; const float* bias2;
  int flags;
  float* C1; int64_t ldc1; int N0;       // optional split of the N range (C1 == nullptr: unused)
  bf16_t* Cb = nullptr; int64_t ldcb = 0; // optional bf16 shadow of C (plain stores only)
  // optional tanh-backward fusion (decoder BPTT, model.lua:649,654-657): x <- (x + dg[m][n]) * (1 - dout[m][n]^2)
  const float* dg = nullptr; const float* dout = nullptr; int64_t ldd = 0;
  // optional (round 4; data gradient in front of a BatchNorm backward, staged 256 x 256 fp32 tiles only: tile256_store_f32): the per-tile partial sums of the
  // BatchNorm backward pass -- (sum d, sum d xhat) per column with d = C[m][n] (y[m][n] > 0), xhat = (x[m][n] - mean[n]) invstd[n] -- in the chunk layout
  // bn_bwd_finalize_kernel / bn_sums_kernel read ([row tile][N][2] doubles), so bn_relu_backward skips its sums pass (a re-read of this output, x and the mask)
  const float* bnb_x = nullptr; const bf16_t* bnb_yb = nullptr; const float* bnb_save = nullptr; double* bnb_part = nullptr;
  template <int NT> __device__ __forceinline__ void quad(int m, int n, int nstep, const float (&v)[NT][4]) const {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      int col = n + nstep * ni;
      if (col >= N) continue;
      float bb = 0.f;
      if (bias) bb = bias[col];
      if (bias2) bb += bias2[col];
      float* base; int64_t ld; int cc;
      if (C1 && col >= N0) { base = C1; ld = ldc1; cc = col - N0; } else { base = C; ld = ldc; cc = col; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int row = m + i;
        if (row >= M) continue;
        float x = v[ni][i] + bb;
        if (flags & EP_RELU) x = fmaxf(x, 0.f);
        if (flags & EP_TANH) x = tanhf_(x);
        if (dg) { const float o = dout[(int64_t)row * ldd + col]; x = (x + dg[(int64_t)row * ldd + col]) * (1.f - o * o); }
        float* p = base + (int64_t)row * ld + cc;
        if (flags & EP_ATOMIC) atomicAdd(p, x);
        else if (flags & EP_ACCUM) *p += x;
        else { *p = x; if (Cb && base == C) Cb[(int64_t)row * ldcb + cc] = (bf16_t)x; }
      }
    }
  }
  // single-element form used by the recurrent-step kernel; prefetch() is issued BEFORE the K loop so that the
  // epilogue's own operand loads overlap it instead of adding a second memory round trip after it
  struct Pre { float dg, dout, b, b2, old; };
  // loads only (see EpGatesFwd::prefetch): the tanh-backward operands, the biases and the value an accumulating store adds to are requested before the K loop
  // instead of inside elem(), where each was a dependent round trip per element (round 6)
  __device__ __forceinline__ Pre prefetch(int row, int n) const {
    Pre p; p.dg = p.dout = p.b = p.b2 = p.old = 0.f;
    const int rr = max(min(row, M - 1), 0), col = min(n, N - 1);
    if (bias) p.b = bias[col];
    if (bias2) p.b2 = bias2[col];
    if (dg) { p.dg = dg[(int64_t)rr * ldd + col]; p.dout = dout[(int64_t)rr * ldd + col]; }
    if (flags & EP_ACCUM) p.old = *((C1 && col >= N0) ? C1 + (int64_t)rr * ldc1 + (col - N0) : C + (int64_t)rr * ldc + col);
    return p;
  }
  __device__ __forceinline__ void prefetch_zx(Pre&, int, int) const {}
  static constexpr bool kCell4 = false; struct Pre4 {};             // (no four-unit form: stepl.h)
  __device__ __forceinline__ bool cell4_ok() const { return false; }
  template <int NT> __device__ __forceinline__ void elem(int row, int n, int nstep, const float (&v)[NT], const Pre& pre) const {
    if (row >= M) return;
    if constexpr (NT == 1) {                                    // the prefetched form (every step kernel calls it per column tile)
      const int col = n;
      if (col >= N) return;
      float x = v[0];
      if (bias) x += pre.b;
      if (bias2) x += pre.b2;
      if (flags & EP_RELU) x = fmaxf(x, 0.f);
      if (flags & EP_TANH) x = tanhf_(x);
      if (dg) x = (x + pre.dg) * (1.f - pre.dout * pre.dout);
      float* p = (C1 && col >= N0) ? C1 + (int64_t)row * ldc1 + (col - N0) : C + (int64_t)row * ldc + col;
      if (flags & EP_ATOMIC) atomicAdd(p, x);
      else if (flags & EP_ACCUM) *p = pre.old + x;
      else { *p = x; if (Cb && !(C1 && col >= N0)) Cb[(int64_t)row * ldcb + col] = (bf16_t)x; }
      return;
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      int col = n + nstep * ni;
      if (col >= N) continue;
      float x = v[ni];
      if (bias) x += bias[col];
      if (bias2) x += bias2[col];
      if (flags & EP_RELU) x = fmaxf(x, 0.f);
      if (flags & EP_TANH) x = tanhf_(x);
      if (dg) { const float o = dout[(int64_t)row * ldd + col]; x = (x + dg[(int64_t)row * ldd + col]) * (1.f - o * o); }
      float* p = (C1 && col >= N0) ? C1 + (int64_t)row * ldc1 + (col - N0) : C + (int64_t)row * ldc + col;
      if (flags & EP_ATOMIC) atomicAdd(p, x);
      else if (flags & EP_ACCUM) *p += x;
      else { *p = x; if (Cb && !(C1 && col >= N0)) Cb[(int64_t)row * ldcb + col] = (bf16_t)x; }
    }
  }
};

// The fp32 tile of gemm_dma_narrow_kernel (256 x 64 NT, 8 waves as 4 (M) x 2 (N), wave tile 64 x 32 NT) through LDS, two passes of 128 rows.
// Round 4: the hoisted bf16 GEMMs (K = 512 / 1024: 16-32 steps, 2 us of MFMA per workgroup) are epilogue-bound -- the quad epilogue issues 64
// 4-byte store instructions per wave for a 128 KB tile; staged, the tile leaves as 16-byte stores of whole 512-byte rows (and its bf16 shadow, when
// asked for, as 8-byte stores from the same image).  Plain stores only (bias / ReLU / tanh applied on the way in); anything else -> quad epilogue.
template <int NT, class EPT>
__device__ __forceinline__ bool narrow_store_staged(const EPT& ep, const f32x16 (&acc)[2][NT], unsigned char* lds, int m_blk, int n_blk, int wm, int wn, int r, int h, int tid) {
  constexpr int BN = 64 * NT, PITCH = BN * 4;
  if ((ep.flags & (EP_ATOMIC | EP_ACCUM)) || ep.C1 || ep.dg || n_blk + BN > ep.N) return false;
  float bb[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) { const int col = n_blk + wn * 32 * NT + ni * 32 + r; bb[ni] = (ep.bias ? ep.bias[col] : 0.f) + (ep.bias2 ? ep.bias2[col] : 0.f); }
  __syncthreads();                                        // every wave is out of the K loop (the ring is free)
  for (int p = 0; p < 2; ++p) {
    if ((wm >> 1) == p) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float x = acc[mi][ni][4 * q + i] + bb[ni];
              if (ep.flags & EP_RELU) x = fmaxf(x, 0.f);
              if (ep.flags & EP_TANH) x = tanhf_(x);
              *reinterpret_cast<float*>(lds + ((wm & 1) * 64 + mi * 32 + 8 * q + 4 * h + i) * PITCH + (wn * 32 * NT + ni * 32 + r) * 4) = x;
            }
    }
    __syncthreads();
    const int row_base = m_blk + p * 128;
#pragma unroll 4
    for (int it = 0; it < (128 * BN / 4) / 512; ++it) {
      const int idx = it * 512 + tid, row = idx / (BN / 4), c = idx % (BN / 4);
      if (row_base + row < ep.M) {
        const float4 v = *reinterpret_cast<const float4*>(lds + row * PITCH + c * 16);
        if (ep.C) *reinterpret_cast<float4*>(ep.C + (int64_t)(row_base + row) * ep.ldc + n_blk + c * 4) = v;      // (C == nullptr: only the bf16 shadow is wanted -- gemm_hh_shadow)
        if (ep.Cb) {
          typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
          bf16x4_ o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
          *reinterpret_cast<bf16x4_*>(ep.Cb + (int64_t)(row_base + row) * ep.ldcb + n_blk + c * 4) = o;
        }
      }
    }
    __syncthreads();
  }
  return true;
}

// conv epilogue: bias, optional ReLU, optional fused max-pool (window = consecutive rows).
struct EpConv {
  float* y; uint8_t* idx; const float* bias; int Cout; int rows; int pmode; int relu;   // y may be null when yb is given (bf16 mode keeps only the shadow)
  bf16_t* yb;                              // optional bf16 shadow of y (operand of the next contraction)
  // optional evaluation-mode BatchNorm + ReLU folded into the store (cnn.lua:23,32 under evaluate(): a per-channel affine map of the
  // conv output): bn_save = {mean, 1/sqrt(var + eps)}[2][Cout] (bn_eval_prepare), the same expression as bn_apply_relu_kernel
  const float* bn_save = nullptr; const float* bn_w = nullptr; const float* bn_b = nullptr;
  // optional (training-mode BatchNorm behind this conv, staged 256 x 256 tiles only: tile256_store_f32): per-tile partial sums (sum y, sum y^2)
  // of every output column, [row tile][Cout][2] doubles -- the layout bn_fwd_finalize_kernel / bn_sums_kernel read, so the statistics pass over y is not needed
  double* bn_part = nullptr;
  // with bn_part: write the (pre-BatchNorm) output as bf16 here instead of fp32 into y -- its statistics are taken from the fp32 accumulators in the
  // epilogue, and every later reader (apply pass, both backward passes) then moves half the bytes (staged tiles only: conv_forward guarantees that)
  bf16_t* y16 = nullptr;
  template <int NT> __device__ __forceinline__ void quad(int m, int n, int nstep, const float (&v)[NT][4]) const {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      int col = n + nstep * ni;
      if (col >= Cout) continue;
      float bb = bias ? bias[col] : 0.f;
      float x[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[i] = v[ni][i] + bb; if (relu) x[i] = fmaxf(x[i], 0.f); }
      if (bn_save) {
        const float mu = bn_save[col], iv = bn_save[Cout + col], ww = bn_w[col], b2 = bn_b[col];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = fmaxf((x[i] - mu) * iv * ww + b2, 0.f);
      }
      if (pmode == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) if (m + i < rows) {
          if (y) y[(int64_t)(m + i) * Cout + col] = x[i];
          if (yb) yb[(int64_t)(m + i) * Cout + col] = (bf16_t)x[i];
        }
      } else if (pmode == 1) {
        if (m < rows) {
          float best = x[0]; int bi = 0;
#pragma unroll
          for (int i = 1; i < 4; ++i) if (x[i] > best) { best = x[i]; bi = i; }
          int64_t o = (int64_t)(m >> 2) * Cout + col;
          if (y) y[o] = best;
          if (idx) idx[o] = (uint8_t)bi; if (yb) yb[o] = (bf16_t)best;
        }
      } else {
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          int row = m + 2 * w;
          if (row < rows) {
            float a = x[2 * w], b = x[2 * w + 1];
            int64_t o = (int64_t)(row >> 1) * Cout + col;
            if (y) y[o] = (b > a) ? b : a;
            if (idx) idx[o] = (uint8_t)(b > a); if (yb) yb[o] = (bf16_t)((b > a) ? b : a);
          }
        }
      }
    }
  }
};

// LSTM cell forward, LSTM.lua:79-105: gates [in, forget, out, g]; v[g][i] = recurrent/input GEMM part.
struct EpGatesFwd {
  const float* zx; int64_t ldzx;          // optional pre-computed input part [m][g*H+j] (biases included)
  const float* b1; const float* b2;       // optional biases [4H]
  const float* c_prev; int64_t ldcp;
  float* c_out; int64_t ldc;
  float* h_out; int64_t ldh;
  float* h_out2; int64_t ldh2;            // optional second copy of h (context slice / attention concat)
  float* gates; int64_t ldg;              // optional, post-activation [m][g*H+j]
  int M, H;
  bf16_t* hb = nullptr; int64_t ldhb = 0;   // optional bf16 shadows of h_out / h_out2 (operands of the next contractions)
  bf16_t* hb2 = nullptr; int64_t ldhb2 = 0;
  const int32_t* zx_tok = nullptr; int64_t zx_tok_stride = 1;   // optional: zx is a per-token table, row r reads table row zx_tok[r*stride]-1
  DropSpec drop;                            // dropout of the SECOND copy (h_out2 / hb2 = the next layer's input), idx = row * H + j
  __device__ __forceinline__ int64_t zrow(int row) const { return zx_tok ? (int64_t)(zx_tok[(int64_t)row * zx_tok_stride] - 1) : (int64_t)row; }
  template <int NT> __device__ __forceinline__ void quad(int m, int j, int, const float (&v)[NT][4]) const {
    static_assert(NT == 4, "gate epilogue needs the 4 gate tiles");
    if (j >= H) return;
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (b1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) bb[g] = b1[g * H + j] + b2[g * H + j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = m + i;
      if (row >= M) continue;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        z[g] = v[g][i] + bb[g];
        if (zx) z[g] += zx[zrow(row) * ldzx + g * H + j];
      }
      float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), og = sigmoidf_(z[2]), gg = tanhf_(z[3]);
      float c = fg * c_prev[(int64_t)row * ldcp + j] + ig * gg;
      float hh = og * tanhf_(c);
      c_out[(int64_t)row * ldc + j] = c;
      h_out[(int64_t)row * ldh + j] = hh;
      const float hh2 = drop.on() ? hh * drop.mask((long long)row * H + j) : hh;
      if (h_out2) h_out2[(int64_t)row * ldh2 + j] = hh2;
      if (hb) hb[(int64_t)row * ldhb + j] = (bf16_t)hh;
      if (hb2) hb2[(int64_t)row * ldhb2 + j] = (bf16_t)hh2;
      if (gates) {
        float* gp = gates + (int64_t)row * ldg + j;
        gp[0] = ig; gp[H] = fg; gp[2 * H] = og; gp[3 * H] = gg;
      }
    }
  }
  // The epilogue's own operands, requested BEFORE the K loop.  prefetch() only LOADS (round 6): nothing here consumes a loaded value -- no sum of the two
  // biases, no `+= zx`, no early return around the loads -- because every such use is an s_waitcnt on the spot, and the 8-16 prefetch calls of a thread then
  // became 8-16 serial memory round trips in front of the K loop (13-28 us of the 31-44 us of a gate launch at 400 rows, tools/ubench/step400.hip).  Absent
  // operands and rows / columns past the end load a valid dummy element instead (the row's c_prev) and are dropped in elem(); the sums are formed there, in the
  // order they always had: z = v + ((b1 + b2) + zx).
  struct Pre { float zx[4]; float b1[4]; float b2[4]; float cp; int32_t tok; };
  __device__ __forceinline__ Pre prefetch(int row, int j) const {
    Pre p;
    const int rr = max(min(row, M - 1), 0), jj = min(j, H - 1);
    const float* const cpp = c_prev + (int64_t)rr * ldcp + jj;
    const int32_t* const tp = zx_tok ? zx_tok + (int64_t)rr * zx_tok_stride : reinterpret_cast<const int32_t*>(cpp);
    p.tok = *tp;
    const float* const b1p = b1 ? b1 + jj : cpp; const float* const b2p = b1 ? b2 + jj : cpp; const int bs = b1 ? H : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) { p.b1[g] = b1p[g * bs]; p.b2[g] = b2p[g * bs]; }
    p.cp = *cpp;
    return p;
  }
  // second half of the prefetch: the gate inputs, whose row may be a token looked up by the first half (all first halves are issued before any second half)
  __device__ __forceinline__ void prefetch_zx(Pre& p, int row, int j) const {
    const int rr = max(min(row, M - 1), 0), jj = min(j, H - 1);
    const float* const cpp = c_prev + (int64_t)rr * ldcp + jj;
    const int64_t zr = zx_tok ? (int64_t)(p.tok - 1) : (int64_t)rr;
    const float* const zp = zx ? zx + zr * ldzx + jj : cpp; const int zs = zx ? H : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) p.zx[g] = zp[g * zs];
  }
  template <int NT> __device__ __forceinline__ void elem(int row, int j, int, const float (&v)[NT], const Pre& pre) const {
    static_assert(NT == 4, "gate epilogue needs the 4 gate tiles");
    if (j >= H || row >= M) return;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float zin = 0.f;
      if (b1) zin = pre.b1[g] + pre.b2[g];
      if (zx) zin += pre.zx[g];
      z[g] = v[g] + zin;
    }
    float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), og = sigmoidf_(z[2]), gg = tanhf_(z[3]);
    float c = fg * pre.cp + ig * gg;
    float hh = og * tanhf_(c);
    c_out[(int64_t)row * ldc + j] = c;
    h_out[(int64_t)row * ldh + j] = hh;
    const float hh2 = drop.on() ? hh * drop.mask((long long)row * H + j) : hh;
    if (h_out2) h_out2[(int64_t)row * ldh2 + j] = hh2;
    if (hb) hb[(int64_t)row * ldhb + j] = (bf16_t)hh;
    if (hb2) hb2[(int64_t)row * ldhb2 + j] = (bf16_t)hh2;
    if (gates) {
      float* gp = gates + (int64_t)row * ldg + j;
      gp[0] = ig; gp[H] = fg; gp[2 * H] = og; gp[3 * H] = gg;
    }
  }
  // ---- four consecutive hidden units j .. j+3 of one row per thread (stepl.h): the same arithmetic per unit, 16-byte loads and stores (8-byte for bf16)
  static constexpr bool kCell4 = true;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef bf16_t v4h __attribute__((ext_vector_type(4)));
  struct Pre4 { v4f zx[4]; v4f b1[4]; v4f b2[4]; v4f cp; int32_t tok; };
  __device__ __forceinline__ bool cell4_ok() const {                 // every row start 16-byte aligned (uniform: kernel arguments only)
    auto a16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    auto a8 = [](const void* q) { return ((uintptr_t)q & 7) == 0; };
    return H % 4 == 0 && a16(c_prev) && ldcp % 4 == 0 && a16(c_out) && ldc % 4 == 0 && a16(h_out) && ldh % 4 == 0 && (!zx || (a16(zx) && ldzx % 4 == 0)) &&
           (!b1 || (a16(b1) && a16(b2))) && (!h_out2 || (a16(h_out2) && ldh2 % 4 == 0)) && (!gates || (a16(gates) && ldg % 4 == 0)) &&
           (!hb || (a8(hb) && ldhb % 4 == 0)) && (!hb2 || (a8(hb2) && ldhb2 % 4 == 0));
  }
  __device__ __forceinline__ Pre4 prefetch4(int row, int j) const {      // loads only, as prefetch()
    Pre4 p;
    const int rr = max(min(row, M - 1), 0), jj = min(j, H - 4);
    const float* const cpp = c_prev + (int64_t)rr * ldcp + jj;
    const int32_t* const tp = zx_tok ? zx_tok + (int64_t)rr * zx_tok_stride : reinterpret_cast<const int32_t*>(cpp);
    p.tok = *tp;
    const float* const b1p = b1 ? b1 + jj : cpp; const float* const b2p = b1 ? b2 + jj : cpp; const int bs = b1 ? H : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) { p.b1[g] = *reinterpret_cast<const v4f*>(b1p + g * bs); p.b2[g] = *reinterpret_cast<const v4f*>(b2p + g * bs); }
    p.cp = *reinterpret_cast<const v4f*>(cpp);
    return p;
  }
  __device__ __forceinline__ void prefetch4_zx(Pre4& p, int row, int j) const {
    const int rr = max(min(row, M - 1), 0), jj = min(j, H - 4);
    const float* const cpp = c_prev + (int64_t)rr * ldcp + jj;
    const int64_t zr = zx_tok ? (int64_t)(p.tok - 1) : (int64_t)rr;
    const float* const zp = zx ? zx + zr * ldzx + jj : cpp; const int zs = zx ? H : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) p.zx[g] = *reinterpret_cast<const v4f*>(zp + g * zs);
  }
  __device__ __forceinline__ void cell4(int row, int j, const v4f (&v)[4], const Pre4& pre) const {
    if (j >= H || row >= M) return;
    v4f gate[4], c4, h4, h24;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float zin = 0.f;
        if (b1) zin = pre.b1[g][u] + pre.b2[g][u];
        if (zx) zin += pre.zx[g][u];
        z[g] = v[g][u] + zin;
      }
      const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), og = sigmoidf_(z[2]), gg = tanhf_(z[3]);
      const float c = fg * pre.cp[u] + ig * gg;
      const float hh = og * tanhf_(c);
      gate[0][u] = ig; gate[1][u] = fg; gate[2][u] = og; gate[3][u] = gg; c4[u] = c; h4[u] = hh;
      h24[u] = drop.on() ? hh * drop.mask((long long)row * H + j + u) : hh;
    }
    *reinterpret_cast<v4f*>(c_out + (int64_t)row * ldc + j) = c4;
    *reinterpret_cast<v4f*>(h_out + (int64_t)row * ldh + j) = h4;
    if (h_out2) *reinterpret_cast<v4f*>(h_out2 + (int64_t)row * ldh2 + j) = h24;
    if (hb) { v4h t; t[0] = (bf16_t)h4[0]; t[1] = (bf16_t)h4[1]; t[2] = (bf16_t)h4[2]; t[3] = (bf16_t)h4[3]; *reinterpret_cast<v4h*>(hb + (int64_t)row * ldhb + j) = t; }
    if (hb2) { v4h t; t[0] = (bf16_t)h24[0]; t[1] = (bf16_t)h24[1]; t[2] = (bf16_t)h24[2]; t[3] = (bf16_t)h24[3]; *reinterpret_cast<v4h*>(hb2 + (int64_t)row * ldhb2 + j) = t; }
    if (gates) {
      float* const gp = gates + (int64_t)row * ldg + j;
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<v4f*>(gp + (int64_t)g * H) = gate[g];
    }
  }
};

// LSTM cell backward: v[0][i] = GEMM part of d(h_out) (recurrent or from the layer above).
struct EpGatesBwd {
  const float* dh1; int64_t ld1;          // optional extra d(h_out) terms [m][j]
  const float* dh2; int64_t ld2;
  const float* dh3 = nullptr; int64_t ld3 = 0;      // (round 6: a third term -- the attention part of d h_top when the chain scores against ctx W_a)
  const float* dc_in; int64_t lddc;       // optional d(c_out) [m][j]
  const float* gates; int64_t ldg;
  const float* c_prev; int64_t ldcp;
  const float* c; int64_t ldcc;
  float* dz; int64_t lddz;                // [m][g*H+j]
  float* dc_out; int64_t lddco;           // d(c_prev)
  int M, H;
  bf16_t* dzb = nullptr; int64_t lddzb = 0; // optional bf16 shadow of dz
  DropSpec drop;                            // dropout backward of the layer above's input: the GEMM part (d of the masked copy) is scaled by the mask, idx = row * H + j
  bool gil = false;                         // gates stored [m][j][4] (interleaved per unit: the decoder cluster kernel) instead of [m][g*H+j]
  template <int NT> __device__ __forceinline__ void quad(int m, int j, int, const float (&v)[NT][4]) const {
    static_assert(NT == 1, "gate backward epilogue is single-tile");
    if (j >= H) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = m + i;
      if (row >= M) continue;
      float dh = v[0][i];
      if (drop.on()) dh *= drop.mask((long long)row * H + j);
      if (dh1) dh += dh1[(int64_t)row * ld1 + j];
      if (dh2) dh += dh2[(int64_t)row * ld2 + j];
      if (dh3) dh += dh3[(int64_t)row * ld3 + j];
      float ig, fg, og, gg;
      if (gil) { const float4 g4 = *reinterpret_cast<const float4*>(gates + (int64_t)row * ldg + 4 * j); ig = g4.x; fg = g4.y; og = g4.z; gg = g4.w; }
      else { const float* gp = gates + (int64_t)row * ldg + j; ig = gp[0]; fg = gp[H]; og = gp[2 * H]; gg = gp[3 * H]; }
      float tc = tanhf_(c[(int64_t)row * ldcc + j]);
      float dc = dh * og * (1.f - tc * tc);
      if (dc_in) dc += dc_in[(int64_t)row * lddc + j];
      float d_o = dh * tc;
      float di = dc * gg, dg = dc * ig, df = dc * c_prev[(int64_t)row * ldcp + j];
      float* zp = dz + (int64_t)row * lddz + j;
      const float z0 = di * ig * (1.f - ig), z1 = df * fg * (1.f - fg), z2 = d_o * og * (1.f - og), z3 = dg * (1.f - gg * gg);
      zp[0] = z0; zp[H] = z1; zp[2 * H] = z2; zp[3 * H] = z3;
      if (dzb) {
        bf16_t* zb = dzb + (int64_t)row * lddzb + j;
        zb[0] = (bf16_t)z0; zb[H] = (bf16_t)z1; zb[2 * H] = (bf16_t)z2; zb[3 * H] = (bf16_t)z3;
      }
      dc_out[(int64_t)row * lddco + j] = dc * fg;
    }
  }
  // loads only, see EpGatesFwd::prefetch: absent terms and rows / columns past the end load the row's c (dropped in elem())
  struct Pre { float dh1, dh2, dh3, dc, ig, fg, og, gg, c, cp; };
  __device__ __forceinline__ Pre prefetch(int row, int j) const {
    Pre p;
    const int rr = max(min(row, M - 1), 0), jj = min(j, H - 1);
    const float* const cq = c + (int64_t)rr * ldcc + jj;
    p.dh1 = *(dh1 ? dh1 + (int64_t)rr * ld1 + jj : cq);
    p.dh2 = *(dh2 ? dh2 + (int64_t)rr * ld2 + jj : cq);
    p.dh3 = *(dh3 ? dh3 + (int64_t)rr * ld3 + jj : cq);
    p.dc = *(dc_in ? dc_in + (int64_t)rr * lddc + jj : cq);
    if (gil) { const float4 g4 = *reinterpret_cast<const float4*>(gates + (int64_t)rr * ldg + 4 * jj); p.ig = g4.x; p.fg = g4.y; p.og = g4.z; p.gg = g4.w; }
    else { const float* gp = gates + (int64_t)rr * ldg + jj; p.ig = gp[0]; p.fg = gp[H]; p.og = gp[2 * H]; p.gg = gp[3 * H]; }
    p.c = *cq; p.cp = c_prev[(int64_t)rr * ldcp + jj];
    return p;
  }
  __device__ __forceinline__ void prefetch_zx(Pre&, int, int) const {}
  static constexpr bool kCell4 = false; struct Pre4 {};
  __device__ __forceinline__ bool cell4_ok() const { return false; }
  template <int NT> __device__ __forceinline__ void elem(int row, int j, int, const float (&v)[NT], const Pre& pre) const {
    static_assert(NT == 1, "gate backward epilogue is single-tile");
    if (j >= H || row >= M) return;
    float pdh = 0.f;
    if (dh1) pdh = pre.dh1;
    if (dh2) pdh += pre.dh2;
    if (dh3) pdh += pre.dh3;
    const float pdc = dc_in ? pre.dc : 0.f;
    float dh = (drop.on() ? v[0] * drop.mask((long long)row * H + j) : v[0]) + pdh;
    float ig = pre.ig, fg = pre.fg, og = pre.og, gg = pre.gg;
    float tc = tanhf_(pre.c);
    float dc = dh * og * (1.f - tc * tc) + pdc;
    float d_o = dh * tc;
    float di = dc * gg, dg = dc * ig, df = dc * pre.cp;
    const float z0 = di * ig * (1.f - ig), z1 = df * fg * (1.f - fg), z2 = d_o * og * (1.f - og), z3 = dg * (1.f - gg * gg);
    float* zp = dz + (int64_t)row * lddz + j;
    zp[0] = z0; zp[H] = z1; zp[2 * H] = z2; zp[3 * H] = z3;
    if (dzb) {
      bf16_t* zb = dzb + (int64_t)row * lddzb + j;
      zb[0] = (bf16_t)z0; zb[H] = (bf16_t)z1; zb[2 * H] = (bf16_t)z2; zb[3 * H] = (bf16_t)z3;
    }
    dc_out[(int64_t)row * lddco + j] = dc * fg;
  }
};

}  // namespace aocr
