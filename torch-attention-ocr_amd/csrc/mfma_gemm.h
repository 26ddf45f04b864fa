// mfma_gemm.h -- MFMA matrix-product core shared by every contraction on the hot path.
//
// Design (gfx950 / CDNA4, 64-lane waves):
//   * operands go global -> VGPR directly in MFMA fragment order (no LDS stage): the
//     fp32-input MFMA (v_mfma_f32_32x32x2_f32) retires 4096 FLOP per 64 cycles, so one
//     16-byte load per lane feeds four MFMAs; L1/L2 carry the small reuse between waves.
//   * the two k-slots of a 32x32x2 MFMA are filled from a float4 (k = kb+4*half+s for
//     step s): any k assignment is legal as long as A and B agree, and this one makes
//     the K-contiguous operand a single dwordx4 load per 8-deep chunk.
//   * bf16 mode converts the same fp32 operands to bf16 in registers
//     (v_cvt_pk_bf16_f32) and issues v_mfma_f32_32x32x16_bf16 on 16-deep chunks.
//   * "loaders" describe an operand (plain row-major, column-major, implicit im2col
//     views of a channels-last tensor); "epilogues" consume the accumulator in quads of
//     4 consecutive rows x 1 column (the 32x32 C/D layout puts rows 4h..4h+3 of a column
//     in 4 consecutive registers), which is exactly a 2x2 / 2x1 max-pool window or the
//     four LSTM gates of one hidden unit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace aocr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one operand fragment of a 32-row tile for one k-chunk: 4 floats (f32 mode, chunk 8)
// or 8 floats (bf16 mode, chunk 16) per lane.
template <int NV> struct Frag { float v[NV]; };

template <bool BF16> struct Mode {
  static constexpr int NV = BF16 ? 8 : 4;       // floats per lane per chunk
  static constexpr int CHUNK = BF16 ? 16 : 8;   // k per chunk
};

// ---------------------------------------------------------------------------
// loaders.  row(r) -> per-row context; load(frag, ctx, k) fills NV consecutive k
// starting at k (k is a multiple of NV; out-of-range -> 0).
// ---------------------------------------------------------------------------

// K-contiguous: element(r,k) = p[r*ld + k]; optional second K segment (concatenated operands [x0 ; x1]).
struct LoadK {
  const float* p0; int64_t ld0; int K0;
  const float* p1; int64_t ld1;
  int rows; int K; int vec;                       // vec: 16-byte aligned -> dwordx4 loads
  struct Ctx { const float* b0; const float* b1; bool ok; };
  __device__ __forceinline__ Ctx row(int r) const {
    Ctx c; c.ok = r < rows; int rr = c.ok ? r : 0;
    c.b0 = p0 + (int64_t)rr * ld0; c.b1 = p1 ? p1 + (int64_t)rr * ld1 : p0;
    return c;
  }
  template <int NV> __device__ __forceinline__ void load(Frag<NV>& f, const Ctx& c, int k) const {
    const float* src = (k < K0) ? c.b0 + k : c.b1 + (k - K0);
    if (c.ok && k + NV <= K && vec) {
#pragma unroll
      for (int j = 0; j < NV; j += 4) {
        float4 t = *reinterpret_cast<const float4*>(src + j);
        f.v[j] = t.x; f.v[j + 1] = t.y; f.v[j + 2] = t.z; f.v[j + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) f.v[j] = (c.ok && k + j < K) ? src[j] : 0.f;
    }
  }
  // branch-free variant for the recurrent-step kernels: the caller guarantees 16-byte alignment and K, K0 multiples
  // of the chunk; rows past the end were clamped to row 0 by row() and their results are dropped by the epilogue.
  template <int NV> __device__ __forceinline__ void load_fast(Frag<NV>& f, const Ctx& c, int k) const {
    const float* src = (k < K0) ? c.b0 + k : c.b1 + (k - K0);
#pragma unroll
    for (int j = 0; j < NV; j += 4) {
      float4 t = *reinterpret_cast<const float4*>(src + j);
      f.v[j] = t.x; f.v[j + 1] = t.y; f.v[j + 2] = t.z; f.v[j + 3] = t.w;
    }
  }
  // branch-free 8-k chunk interface of gemm_f32t_kernel (host guarantees vec, K % 8 == 0, K0 % 8 == 0): the address of a chunk that is
  // not there is the base pointer (a valid 32 bytes) and `ok` tells the stager to keep zeros instead of what was loaded
  typedef int KCur;
  __device__ __forceinline__ KCur kseek(int k) const { return k; }
  __device__ __forceinline__ void kadvance(KCur& k, int dk) const { k += dk; }
  __device__ __forceinline__ const float* ptr8(const Ctx& c, const KCur& k, bool& ok) const {
    ok = c.ok && k + 8 <= K;
    const float* s = (k < K0) ? c.b0 + k : c.b1 + (k - K0);
    return ok ? s : p0;
  }
};

// MN-contiguous: element(r,k) = p[k*ld + r]
struct LoadMN {
  const float* p; int64_t ld; int rows; int K; int vec;
  struct Ctx { const float* b; bool ok; };
  __device__ __forceinline__ Ctx row(int r) const { Ctx c; c.ok = r < rows; c.b = p + (c.ok ? r : 0); return c; }
  template <int NV> __device__ __forceinline__ void load(Frag<NV>& f, const Ctx& c, int k) const {
#pragma unroll
    for (int j = 0; j < NV; ++j) f.v[j] = (c.ok && k + j < K) ? c.b[(int64_t)(k + j) * ld] : 0.f;
  }
  template <int NV> __device__ __forceinline__ void load_fast(Frag<NV>& f, const Ctx& c, int k) const { load<NV>(f, c, k); }
  // micro-block interface of the LDS-tiled kernel: v[kk][j] = element(r + j, k + kk), 4 rows x 4 k, r % 4 == 0
  struct Ctx4 { const float* b; int nv; };
  __device__ __forceinline__ Ctx4 row4(int r) const {
    Ctx4 c; c.nv = min(max(rows - r, 0), 4); c.b = p + (c.nv > 0 ? r : 0); return c;
  }
  __device__ __forceinline__ void load4x4(float (&v)[4][4], const Ctx4& c, int k) const {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float* s = c.b + (int64_t)(k + kk) * ld;
      if (k + kk < K && c.nv == 4 && vec) {
        float4 t = *reinterpret_cast<const float4*>(s);
        v[kk][0] = t.x; v[kk][1] = t.y; v[kk][2] = t.z; v[kk][3] = t.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[kk][j] = (k + kk < K && j < c.nv) ? s[j] : 0.f;
      }
    }
  }
};

// Implicit im2col of a channels-last tensor src (B,H,W,C), K-contiguous:
//   row m  -> a pixel (b,y,x) of the "row grid" (Hr x Wr), ordered so that a pool window is
//             4 (2x2) or 2 (2x1) consecutive rows when pmode != 0;
//   k      -> (tap, c), tap = kh*KW + kw; element = src[b, y + sgn*kh + off, x + sgn*kw + off, c].
// forward conv: sgn=+1, off=-pad over the input; data-gradient: sgn=-1, off=+pad over dY.
struct LoadConvK {
  const float* src; int H, W, C;
  int KW, sgn, off;
  int Hr, Wr;            // row grid (un-pooled)
  int pmode, Hp, Wp;     // 0 none | 1 2x2 | 2 (kH2,kW1)
  int rows, K;
  struct Ctx { int b, y, x; bool ok; };
  __device__ __forceinline__ Ctx row(int m) const {
    Ctx c; c.ok = m < rows; int mm = c.ok ? m : 0;
    if (pmode == 1) {
      int win = mm >> 2, dy = (mm >> 1) & 1, dx = mm & 1;
      int px = win % Wp; int t = win / Wp; int py = t % Hp; c.b = t / Hp;
      c.y = 2 * py + dy; c.x = 2 * px + dx;
    } else if (pmode == 2) {
      int win = mm >> 1, dy = mm & 1;
      c.x = win % Wr; int t = win / Wr; int py = t % Hp; c.b = t / Hp; c.y = 2 * py + dy;
    } else {
      c.x = mm % Wr; int t = mm / Wr; c.y = t % Hr; c.b = t / Hr;
    }
    return c;
  }
  template <int NV> __device__ __forceinline__ void load(Frag<NV>& f, const Ctx& c, int k) const {
    int tap = k / C; int ch = k - tap * C; int kh = tap / KW; int kw = tap - kh * KW;
    int sy = c.y + sgn * kh + off, sx = c.x + sgn * kw + off;
    bool ok = c.ok && k < K && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
    if (ok) {
      const float* s = src + (((int64_t)c.b * H + sy) * W + sx) * C + ch;
#pragma unroll
      for (int j = 0; j < NV; j += 4) {
        float4 t = *reinterpret_cast<const float4*>(s + j);
        f.v[j] = t.x; f.v[j + 1] = t.y; f.v[j + 2] = t.z; f.v[j + 3] = t.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) f.v[j] = 0.f;
    }
  }
  // branch-free 8-k chunk interface (gemm_f32t_kernel; C % 8 == 0, so a chunk stays inside one tap): the (tap, channel) of the cursor advance
  // by addition instead of two integer divisions per load
  struct KCur { int k, ch, kh, kw; };
  __device__ __forceinline__ KCur kseek(int k) const { KCur q; q.k = k; const int tap = k / C; q.ch = k - tap * C; q.kh = tap / KW; q.kw = tap - q.kh * KW; return q; }
  __device__ __forceinline__ void kadvance(KCur& q, int dk) const {
    q.k += dk; q.ch += dk;                       // dk <= C (host: C >= 32): at most one tap boundary per advance, as selects (no divergent loop)
    const bool wrap = q.ch >= C; q.ch -= wrap ? C : 0;
    const int kw1 = q.kw + (wrap ? 1 : 0); const bool row = kw1 == KW;
    q.kw = row ? 0 : kw1; q.kh += row ? 1 : 0;
  }
  __device__ __forceinline__ const float* ptr8(const Ctx& c, const KCur& q, bool& ok) const {
    const int sy = c.y + sgn * q.kh + off, sx = c.x + sgn * q.kw + off;
    ok = c.ok && q.k < K && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
    return ok ? src + (((int64_t)c.b * H + sy) * W + sx) * C + q.ch : src;
  }
};

// B operand of the data-gradient: element(n=ci, k=(tap,co)) = w[co][tap][ci], w stored [Cout][KK][Cin].
struct LoadConvWT {
  const float* w; int Cin, Cout, KK; int K;
  struct Ctx { int ci; bool ok; };
  __device__ __forceinline__ Ctx row(int r) const { Ctx c; c.ok = r < Cin; c.ci = c.ok ? r : 0; return c; }
  template <int NV> __device__ __forceinline__ void load(Frag<NV>& f, const Ctx& c, int k) const {
    int tap = k / Cout; int co = k - tap * Cout;       // NV consecutive k stay inside one tap (Cout % 16 == 0)
#pragma unroll
    for (int j = 0; j < NV; ++j)
      f.v[j] = (c.ok && k + j < K) ? w[((int64_t)(co + j) * KK + tap) * Cin + c.ci] : 0.f;
  }
  struct Ctx4 { int ci; bool ok; };
  __device__ __forceinline__ Ctx4 row4(int r) const { Ctx4 c; c.ok = r + 3 < Cin; c.ci = c.ok ? r : 0; return c; }   // Cin % 4 == 0
  __device__ __forceinline__ void load4x4(float (&v)[4][4], const Ctx4& c, int k) const {
    int tap = k / Cout; int co = k - tap * Cout;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (c.ok && k + kk < K) {
        float4 t = *reinterpret_cast<const float4*>(w + ((int64_t)(co + kk) * KK + tap) * Cin + c.ci);
        v[kk][0] = t.x; v[kk][1] = t.y; v[kk][2] = t.z; v[kk][3] = t.w;
      } else { v[kk][0] = v[kk][1] = v[kk][2] = v[kk][3] = 0.f; }
    }
  }
};

// B operand of the filter-gradient: element(n=(tap,ci), k=pixel p of the output grid) =
//   x[b, y + kh - pad, x + kw - pad, ci]; x (B,H,W,Cin), output grid Ho x Wo in plain order.
struct LoadConvXcol {
  const float* x; int H, W, Cin; int KW, pad; int Ho, Wo; int N; int K;
  struct Ctx { int kh, kw, ci; bool ok; };
  __device__ __forceinline__ Ctx row(int n) const {
    Ctx c; c.ok = n < N; int nn = c.ok ? n : 0; int tap = nn / Cin; c.ci = nn - tap * Cin;
    c.kh = tap / KW; c.kw = tap - c.kh * KW; return c;
  }
  template <int NV> __device__ __forceinline__ void load(Frag<NV>& f, const Ctx& c, int k) const {
    int px = k % Wo; int t = k / Wo; int py = t % Ho; int b = t / Ho;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      int sy = py + c.kh - pad, sx = px + c.kw - pad;
      bool ok = c.ok && (k + j) < K && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
      f.v[j] = ok ? x[(((int64_t)b * H + sy) * W + sx) * Cin + c.ci] : 0.f;
      if (++px == Wo) { px = 0; if (++py == Ho) { py = 0; ++b; } }
    }
  }
  struct Ctx4 { int kh, kw, ci; bool ok; };
  __device__ __forceinline__ Ctx4 row4(int n) const {                     // 4 consecutive ci of one tap (Cin % 4 == 0)
    Ctx4 c; c.ok = n + 3 < N; int nn = c.ok ? n : 0; int tap = nn / Cin; c.ci = nn - tap * Cin;
    c.kh = tap / KW; c.kw = tap - c.kh * KW; return c;
  }
  __device__ __forceinline__ void load4x4(float (&v)[4][4], const Ctx4& c, int k) const {
    int px = k % Wo; int t = k / Wo; int py = t % Ho; int b = t / Ho;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int sy = py + c.kh - pad, sx = px + c.kw - pad;
      bool ok = c.ok && (k + kk) < K && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
      if (ok) {
        float4 tt = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + sy) * W + sx) * Cin + c.ci);
        v[kk][0] = tt.x; v[kk][1] = tt.y; v[kk][2] = tt.z; v[kk][3] = tt.w;
      } else { v[kk][0] = v[kk][1] = v[kk][2] = v[kk][3] = 0.f; }
      if (++px == Wo) { px = 0; if (++py == Ho) { py = 0; ++b; } }
    }
  }
};

// ---------------------------------------------------------------------------
// MFMA step over one chunk
// ---------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 to_bf16x8(const Frag<8>& f) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)f.v[j];
  return v;
}
__device__ __forceinline__ bf16x8 to_bf16x8(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ bf16x8 to_bf16x8(const Frag<4>&) { return bf16x8{}; }     // never used (fp32 mode)
__device__ __forceinline__ float frag_elem(const Frag<4>& f, int s) { return f.v[s]; }
__device__ __forceinline__ float frag_elem(const Frag<8>&, int) { return 0.f; }       // never used (bf16 mode)
__device__ __forceinline__ float frag_elem(const uint4&, int) { return 0.f; }

template <bool BF16, int WM, int WN, class FA, class FB>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[WM][WN], const FA (&a)[WM], const FB (&b)[WN]) {
  if constexpr (BF16) {
    bf16x8 ab[WM], bb[WN];
#pragma unroll
    for (int i = 0; i < WM; ++i) ab[i] = to_bf16x8(a[i]);
#pragma unroll
    for (int i = 0; i < WN; ++i) bb[i] = to_bf16x8(b[i]);
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
      for (int ni = 0; ni < WN; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[mi], bb[ni], acc[mi][ni], 0, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag_elem(a[mi], s), frag_elem(b[ni], s), acc[mi][ni], 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------
// "big" kernel: 4 waves as 2x2, each wave a (32*WM) x (32*WN) tile.
// grid = (ceil(N / (64*WN)), ceil(M / (64*WM)), ksplit); blockIdx.z owns k range
// [z*kper, min(K,(z+1)*kper)).  Epilogue must be atomic/accumulating when ksplit > 1.
// ---------------------------------------------------------------------------
template <bool BF16, int WM, int WN, class AL, class BL, class EP>
__global__ __launch_bounds__(256) void gemm_big_kernel(AL a, BL b, EP ep, int K, int kper) {
  constexpr int NV = Mode<BF16>::NV, CH = Mode<BF16>::CHUNK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * (64 * WM) + (wave >> 1) * (32 * WM);
  const int n0 = blockIdx.x * (64 * WN) + (wave & 1) * (32 * WN);
  const int kbeg = blockIdx.z * kper;
  const int kend = min(K, kbeg + kper);

  typename AL::Ctx ca[WM]; typename BL::Ctx cb[WN];
#pragma unroll
  for (int i = 0; i < WM; ++i) ca[i] = a.row(m0 + 32 * i + r);
#pragma unroll
  for (int i = 0; i < WN; ++i) cb[i] = b.row(n0 + 32 * i + r);

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  Frag<NV> fa[WM], fb[WN], ga[WM], gb[WN];
  if (kbeg < kend) {
#pragma unroll
    for (int i = 0; i < WM; ++i) a.template load<NV>(fa[i], ca[i], kbeg + NV * h);
#pragma unroll
    for (int i = 0; i < WN; ++i) b.template load<NV>(fb[i], cb[i], kbeg + NV * h);
  }
  for (int kb = kbeg; kb < kend; kb += CH) {
    const int kn = kb + CH;
    if (kn < kend) {
#pragma unroll
      for (int i = 0; i < WM; ++i) a.template load<NV>(ga[i], ca[i], kn + NV * h);
#pragma unroll
      for (int i = 0; i < WN; ++i) b.template load<NV>(gb[i], cb[i], kn + NV * h);
    }
    mma_chunk<BF16, WM, WN, Frag<NV>, Frag<NV>>(acc, fa, fb);
#pragma unroll
    for (int i = 0; i < WM; ++i) fa[i] = ga[i];
#pragma unroll
    for (int i = 0; i < WN; ++i) fb[i] = gb[i];
  }
#pragma unroll
  for (int mi = 0; mi < WM; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[WN][4];
#pragma unroll
      for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<WN>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}

// ---------------------------------------------------------------------------
// LDS-tiled bf16 kernel for the large contractions (convolutions, hoisted projections, weight gradients).
// 128 x 128 x 32 block tile, 4 waves as 2x2 (64x64 each = 2x2 MFMA 32x32x16 tiles), two LDS buffers:
//   global (fp32, any loader) -> registers (issued before the MFMAs of the current tile: T14 split)
//   -> v_cvt_pk_bf16_f32 -> ds_write_b128 into the other buffer after the MFMAs -> one barrier per k-step.
// LDS image per operand: [128 rows][32 k] bf16 with an 80-byte row pitch: 20 dwords/row makes any 16 rows that
// are distinct mod 16 hit 16 different 4-bank slots, so the ds_read_b128 fragment reads are conflict-free.
// Blocks are renumbered so that the tiles of one M row-panel (which share the A operand) run on one XCD's L2.
// ---------------------------------------------------------------------------
// ---- operands that already live in HBM as bf16 ("shadow" copies written by the producing kernels): half the
// L2->L1 bytes of the fp32 loaders and no conversion work.  Only the LDS-tiled kernel reads them.
typedef __bf16 bf16_t;

// Each loader keeps a cursor that the stager advances by 32 k per tile, so the steady state has no integer division.

// pointer select without control flow (hipcc otherwise branches around the address arithmetic, which splits the K loop of
// the LDS-DMA kernel into several basic blocks and makes its waitcnt insertion conservative)
__device__ __forceinline__ const bf16_t* dma_select(bool ok, const bf16_t* p, const bf16_t* zero) {
  const uint64_t m = ok ? ~0ull : 0ull;
  return reinterpret_cast<const bf16_t*>((reinterpret_cast<uint64_t>(p) & m) | (reinterpret_cast<uint64_t>(zero) & ~m));
}
// K-contiguous bf16: element(r,k) = p[r*ld + k]; 16-byte aligned rows (ld % 8 == 0).
struct LoadKh {
  const bf16_t* p; int64_t ld; int rows; int K;
  struct Ctx { const bf16_t* b; bool ok; };
  struct Cur { int k; };
  __device__ __forceinline__ Ctx row(int r) const { Ctx c; c.ok = r < rows; c.b = p + (int64_t)(c.ok ? r : 0) * ld; return c; }
  __device__ __forceinline__ Cur seek(int k) const { Cur c; c.k = k; return c; }
  __device__ __forceinline__ void advance(Cur& c) const { c.k += 32; }
  __device__ __forceinline__ uint4 load8(const Ctx& c, const Cur& u, int chunk) const {
    int k = u.k + 8 * chunk;
    if (c.ok && k + 8 <= K) return *reinterpret_cast<const uint4*>(c.b + k);
    return make_uint4(0, 0, 0, 0);
  }
  // LDS-DMA staging (gemm_dma_bf16_kernel): per-row source pointer of this lane's 16-byte chunk, 32 k per tile
  struct DRow { const bf16_t* b; };                   // nullptr: row outside the operand -> zero page
  struct DCur { int k; };
  __device__ __forceinline__ DRow drow(int r, int chunk) const { DRow d; d.b = r < rows ? p + (int64_t)r * ld + 8 * chunk : nullptr; return d; }
  __device__ __forceinline__ DCur dseek(int k) const { DCur c; c.k = k; return c; }
  __device__ __forceinline__ void dadvance(DCur& c) const { c.k += 32; }
  __device__ __forceinline__ const bf16_t* dsrc(const DRow& d, const DCur& u, const bf16_t* zero) const {
    return dma_select(d.b != nullptr && u.k < K, d.b + u.k, zero);
  }
};
// K-contiguous bf16 whose K range is the CONCATENATION of two buffers with the same rows (LDS-DMA staging only): element(r, k) = k < K0 ? p0[r ld0 + k] :
// p1[r ld1 + k - K0], K0 % 32 == 0 (a 32-k tile never straddles).  One product over [d z_fw | d z_bw] x [W_fw ; W_bw] instead of two launches whose
// second re-reads and adds to the first's output (encoder_backward: d X = d z_fw W_i2h_fw + d z_bw W_i2h_bw, model.lua:675,689).
struct LoadKhCat {
  const bf16_t* p0; const bf16_t* p1; int64_t ld0, ld1; int rows; int K0, K;
  struct DRow { const bf16_t* b0; const bf16_t* b1; };
  struct DCur { int k; };
  __device__ __forceinline__ DRow drow(int r, int chunk) const {
    DRow d; const bool ok = r < rows;
    d.b0 = ok ? p0 + (int64_t)r * ld0 + 8 * chunk : nullptr; d.b1 = ok ? p1 + (int64_t)r * ld1 + 8 * chunk : nullptr; return d;
  }
  __device__ __forceinline__ DCur dseek(int k) const { DCur c; c.k = k; return c; }
  __device__ __forceinline__ void dadvance(DCur& c) const { c.k += 32; }
  __device__ __forceinline__ const bf16_t* dsrc(const DRow& d, const DCur& u, const bf16_t* zero) const {
    const bf16_t* b = u.k >= K0 ? d.b1 + (u.k - K0) : d.b0 + u.k;           // (u.k is wave-uniform)
    return dma_select(d.b0 != nullptr && u.k < K, b, zero);
  }
};
// implicit im2col over a bf16 channels-last tensor (same geometry as LoadConvK); cursor = (kh, kw, first channel) of the tile
struct LoadConvKh {
  const bf16_t* src; LoadConvK g;          // g.src unused
  typedef LoadConvK::Ctx Ctx;
  struct Cur { int kh, kw, ch, k; };
  __device__ __forceinline__ Ctx row(int m) const { return g.row(m); }
  __device__ __forceinline__ Cur seek(int k) const {
    Cur c; int tap = k / g.C; c.ch = k - tap * g.C; c.kh = tap / g.KW; c.kw = tap - c.kh * g.KW; c.k = k; return c;
  }
  __device__ __forceinline__ void advance(Cur& c) const {          // C % 32 == 0: a 32-wide tile never straddles a tap
    c.k += 32; c.ch += 32;
    if (c.ch >= g.C) { c.ch -= g.C; if (++c.kw == g.KW) { c.kw = 0; ++c.kh; } }
  }
  __device__ __forceinline__ uint4 load8(const Ctx& c, const Cur& u, int chunk) const {
    int sy = c.y + g.sgn * u.kh + g.off, sx = c.x + g.sgn * u.kw + g.off;
    bool ok = c.ok && u.k < g.K && (unsigned)sy < (unsigned)g.H && (unsigned)sx < (unsigned)g.W;
    if (ok) return *reinterpret_cast<const uint4*>(src + (((int64_t)c.b * g.H + sy) * g.W + sx) * g.C + u.ch + 8 * chunk);
    return make_uint4(0, 0, 0, 0);
  }
  // LDS-DMA staging: C % 32 == 0, so a 32-k tile is 32 channels of ONE tap and (kh, kw, ch) are wave-uniform scalars;
  // per row only the two bounds compares and a pointer select remain.  Taps outside the map read the zero page.
  struct DRow { const bf16_t* b; int y, x; };         // b = &src[b, y+off, x+off, 8*chunk] (may point outside; never dereferenced then)
  struct DCur { int kh, kw, ch, k; };
  __device__ __forceinline__ DRow drow(int m, int chunk) const {
    Ctx c = g.row(m); DRow d; d.y = c.ok ? c.y + g.off : -(1 << 20); d.x = c.x + g.off;
    d.b = src + (((int64_t)c.b * g.H + (c.y + g.off)) * g.W + (c.x + g.off)) * g.C + 8 * chunk; return d;
  }
  __device__ __forceinline__ DCur dseek(int k) const { DCur c; int tap = k / g.C; c.ch = k - tap * g.C; c.kh = tap / g.KW; c.kw = tap - c.kh * g.KW; c.k = k; return c; }
  __device__ __forceinline__ void dadvance(DCur& c) const {
    c.ch += 32; c.k += 32;                            // branch-free: keeps the K loop of the DMA kernel one basic block
    const int w = c.ch >= g.C; c.ch = w ? 0 : c.ch; c.kw += w;
    const int w2 = c.kw == g.KW; c.kw = w2 ? 0 : c.kw; c.kh += w2;
  }
  __device__ __forceinline__ const bf16_t* dsrc(const DRow& d, const DCur& u, const bf16_t* zero) const {
    int dy = g.sgn * u.kh, dx = g.sgn * u.kw;
    bool ok = u.k < g.K && (unsigned)(d.y + dy) < (unsigned)g.H && (unsigned)(d.x + dx) < (unsigned)g.W;
    return dma_select(ok, d.b + ((int64_t)(dy * g.W + dx) * g.C + u.ch), zero);
  }
};
// M/N-contiguous bf16: element(r,k) = p[k*ld + r]; micro-block = 8 rows x 4 k (4 dwordx4 along the rows)
struct LoadMNh {
  const bf16_t* p; int64_t ld; int rows; int K;
  struct Ctx8 { const bf16_t* b; bool ok; };
  struct Cur { int k; const bf16_t* q; };              // q = &element(r, k): advanced by 32 ld per tile (no multiply per load)
  __device__ __forceinline__ Ctx8 row8(int r) const { Ctx8 c; c.ok = r + 7 < rows; c.b = p + (c.ok ? r : 0); return c; }
  __device__ __forceinline__ Cur seek(const Ctx8& c, int k) const { Cur u; u.k = k; u.q = c.b + (int64_t)k * ld; return u; }
  __device__ __forceinline__ void advance(const Ctx8&, Cur& u) const { u.k += 32; u.q += 32 * ld; }
  __device__ __forceinline__ uint4 load8(const Ctx8& c, const Cur& u) const {        // 8 rows of ONE k
    return (c.ok && u.k < K) ? *reinterpret_cast<const uint4*>(u.q) : make_uint4(0, 0, 0, 0);
  }
  __device__ __forceinline__ void load8x4(uint4 (&v)[4], const Ctx8& c, const Cur& u) const {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      v[kk] = (c.ok && u.k + kk < K) ? *reinterpret_cast<const uint4*>(u.q + kk * ld) : make_uint4(0, 0, 0, 0);
  }
};
// filter-gradient B operand over a bf16 input map (same geometry as LoadConvXcol); 8 consecutive ci of one tap;
// cursor = output pixel (b, py, px) of the micro-block's first k
struct LoadConvXcolh {
  const bf16_t* x; LoadConvXcol g;
  struct Ctx8 { int dy, dx, ci; bool ok; };     // dy = kh - pad, dx = kw - pad
  // p = &x[b, py + dy, px + dx, ci] (may point outside the map: never dereferenced then).  When the output grid is the input grid
  // (Ho == H, Wo == W: every 3 x 3 / pad 1 layer) pixel k IS the raster index of the input pixel, so the address is linear in k and
  // advance() is one pointer add -- the (b, y, x) -> address multiplies per piece and step cost the filter-gradient kernels as
  // many VALU issue cycles as their MFMAs.
  struct Cur { int px, py, b, k; const bf16_t* p; };
  __device__ __forceinline__ Ctx8 row8(int n) const {
    Ctx8 c; c.ok = n + 7 < g.N; int nn = c.ok ? n : 0; int tap = nn / g.Cin; c.ci = nn - tap * g.Cin;
    int kh = tap / g.KW; c.dy = kh - g.pad; c.dx = tap - kh * g.KW - g.pad; return c;
  }
  __device__ __forceinline__ const bf16_t* addr(const Ctx8& c, int b, int py, int px) const {
    return x + (((int64_t)b * g.H + (py + c.dy)) * g.W + (px + c.dx)) * g.Cin + c.ci;
  }
  __device__ __forceinline__ Cur seek(const Ctx8& c, int k) const {
    Cur u; u.k = k; u.px = k % g.Wo; int t = k / g.Wo; u.py = t % g.Ho; u.b = t / g.Ho; u.p = addr(c, u.b, u.py, u.px); return u;
  }
  __device__ __forceinline__ void advance(const Ctx8& c, Cur& u) const {
    u.k += 32; u.px += 32;
    if (g.Wo >= 32) {                                   // at most one row wrap: no per-lane loop (uniform branch)
      const int w = u.px >= g.Wo; u.px -= w ? g.Wo : 0; u.py += w;
      const int w2 = u.py == g.Ho; u.py = w2 ? 0 : u.py; u.b += w2;
    } else {
      while (u.px >= g.Wo) { u.px -= g.Wo; if (++u.py == g.Ho) { u.py = 0; ++u.b; } }
    }
    if (g.Ho == g.H && g.Wo == g.W) u.p += 32 * g.Cin; else u.p = addr(c, u.b, u.py, u.px);
  }
  __device__ __forceinline__ bool inside(const Ctx8& c, const Cur& u) const {
    return c.ok && (unsigned)(u.py + c.dy) < (unsigned)g.H && (unsigned)(u.px + c.dx) < (unsigned)g.W;
  }
  // LDS-DMA staging (conv_wgrad_dma_kernel): source pointer of 8 channels of ONE pixel, zero page outside the map / past kend
  __device__ __forceinline__ const bf16_t* dsrc8(const Ctx8& c, const Cur& u, int kend, const bf16_t* zero) const {
    return dma_select(inside(c, u) && u.k < kend, u.p, zero);
  }
  __device__ __forceinline__ uint4 load8(const Ctx8& c, const Cur& u) const {        // 8 channels of ONE pixel
    return (inside(c, u) && u.k < g.K) ? *reinterpret_cast<const uint4*>(u.p) : make_uint4(0, 0, 0, 0);
  }
  __device__ __forceinline__ void load8x4(uint4 (&v)[4], const Ctx8& c, const Cur& u) const {
    int px = u.px, py = u.py, b = u.b;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      int sy = py + c.dy, sx = px + c.dx;
      bool ok = c.ok && (u.k + kk) < g.K && (unsigned)sy < (unsigned)g.H && (unsigned)sx < (unsigned)g.W;
      v[kk] = ok ? *reinterpret_cast<const uint4*>(x + (((int64_t)b * g.H + sy) * g.W + sx) * g.Cin + c.ci) : make_uint4(0, 0, 0, 0);
      if (++px == g.Wo) { px = 0; if (++py == g.Ho) { py = 0; ++b; } }
    }
  }
};

// K-contiguous bf16 with an optional second K segment: the weight shadows read by the recurrent-step kernels.
struct LoadKh2 {
  const bf16_t* p0; int64_t ld0; int K0;
  const bf16_t* p1; int64_t ld1;
  int rows; int K;
  struct Ctx { const bf16_t* b0; const bf16_t* b1; bool ok; };
  __device__ __forceinline__ Ctx row(int r) const {
    Ctx c; c.ok = r < rows; int rr = c.ok ? r : 0;
    c.b0 = p0 + (int64_t)rr * ld0; c.b1 = p1 ? p1 + (int64_t)rr * ld1 : p0;
    return c;
  }
  template <int NV> __device__ __forceinline__ void load(uint4& f, const Ctx& c, int k) const {
    static_assert(NV == 8, "bf16 weight shadows are only read in bf16 mode");
    const bf16_t* src = (k < K0) ? c.b0 + k : c.b1 + (k - K0);
    f = (c.ok && k + 8 <= K) ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);
  }
  template <int NV> __device__ __forceinline__ void load_fast(uint4& f, const Ctx& c, int k) const {
    f = *reinterpret_cast<const uint4*>((k < K0) ? c.b0 + k : c.b1 + (k - K0));
  }
};
template <class L, int NV> struct FragOf { typedef Frag<NV> type; };
template <int NV> struct FragOf<LoadKh2, NV> { typedef uint4 type; };

template <class L> struct KContig { static constexpr bool v = true; };
template <> struct KContig<LoadMN> { static constexpr bool v = false; };
template <> struct KContig<LoadConvWT> { static constexpr bool v = false; };
template <> struct KContig<LoadConvXcol> { static constexpr bool v = false; };
template <> struct KContig<LoadMNh> { static constexpr bool v = false; };
template <> struct KContig<LoadConvXcolh> { static constexpr bool v = false; };
template <class L> struct SrcBf16 { static constexpr bool v = false; };
template <> struct SrcBf16<LoadKh> { static constexpr bool v = true; };
template <> struct SrcBf16<LoadConvKh> { static constexpr bool v = true; };
template <> struct SrcBf16<LoadMNh> { static constexpr bool v = true; };
template <> struct SrcBf16<LoadConvXcolh> { static constexpr bool v = true; };
template <> struct SrcBf16<LoadKh2> { static constexpr bool v = true; };

constexpr int LDS_PITCH = 80;             // bytes per 32-k row of bf16 (64) + 16 pad
constexpr int LDS_PITCH32 = 144;          // bytes per 32-k row of fp32 (128) + 16 pad: 36 dwords, so 16 rows distinct mod 16 hit 16 different 4-bank slots

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Stages this thread's share of one 128 x 32 operand tile: global fp32 -> registers (load) -> bf16 in LDS (store).
template <class LD, int BK = 32, bool KC = KContig<LD>::v, bool H16 = SrcBf16<LD>::v> struct Stager;

// K-contiguous operand: two (row, 8-k) items per thread, 4 lanes cover one row's 128 bytes.
template <class LD> struct Stager<LD, 32, true, false> {
  int row[2], chunk;
  typename LD::Ctx ctx[2];
  Frag<8> reg[2];
  int k0;
  __device__ __forceinline__ void init(const LD& l, int base, int tid, int, int kbeg) {
    chunk = tid & 3; k0 = kbeg;
#pragma unroll
    for (int i = 0; i < 2; ++i) { row[i] = (tid >> 2) + 64 * i; ctx[i] = l.row(base + row[i]); }
  }
  __device__ __forceinline__ void load(const LD& l) {
#pragma unroll
    for (int i = 0; i < 2; ++i) l.template load<8>(reg[i], ctx[i], k0 + 8 * chunk);
    k0 += 32;
  }
  __device__ __forceinline__ void skip(const LD&) { k0 += 32; }
  __device__ __forceinline__ void store(unsigned char* tile) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)reg[i].v[j];
      *reinterpret_cast<bf16x8*>(tile + row[i] * LDS_PITCH + chunk * 16) = v;
    }
  }
  __device__ __forceinline__ void store32(unsigned char* tile) const {          // exact-fp32 image (gemm_lds_f32_kernel): 144-byte rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4* d = reinterpret_cast<float4*>(tile + row[i] * LDS_PITCH32 + chunk * 32);
      d[0] = make_float4(reg[i].v[0], reg[i].v[1], reg[i].v[2], reg[i].v[3]);
      d[1] = make_float4(reg[i].v[4], reg[i].v[5], reg[i].v[6], reg[i].v[7]);
    }
  }
};

// M/N-contiguous operand: one 4-row x 4-k micro-block per thread, loaded as 4 dwordx4 along the contiguous
// dimension (32 lanes = 512 contiguous bytes) and transposed in registers into four 8-byte LDS writes.
template <class LD> struct Stager<LD, 32, false, false> {
  int row4, k4;
  typename LD::Ctx4 ctx;
  float reg[4][4]; int k0;
  __device__ __forceinline__ void init(const LD& l, int base, int tid, int, int kbeg) {
    k0 = kbeg;
    // lanes run over the 8 k-groups first: the 8-byte transposed LDS writes of 16 consecutive lanes then hit 16
    // different bank pairs (rows 4 apart are 16 banks apart at the 80-byte pitch); each global row still gets
    // whole 128-byte lines from the 8 lanes that share a k.
    k4 = (tid & 7) * 4; row4 = (tid >> 3) * 4; ctx = l.row4(base + row4);
  }
  __device__ __forceinline__ void load(const LD& l) { l.load4x4(reg, ctx, k0 + k4); k0 += 32; }
  __device__ __forceinline__ void skip(const LD&) { k0 += 32; }
  __device__ __forceinline__ void store(unsigned char* tile) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 v;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) v[kk] = (__bf16)reg[kk][j];
      *reinterpret_cast<bf16x4*>(tile + (row4 + j) * LDS_PITCH + k4 * 2) = v;
    }
  }
  __device__ __forceinline__ void store32(unsigned char* tile) const {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(tile + (row4 + j) * LDS_PITCH32 + k4 * 4) = make_float4(reg[0][j], reg[1][j], reg[2][j], reg[3][j]);
  }
};

// bf16 source, K-contiguous: two 16-byte items per thread, copied straight into LDS.
template <class LD, int BK> struct Stager<LD, BK, true, true> {
  static constexpr int CPR = BK / 8, NI = BK / 16, PITCH = BK * 2 + 16;      // chunks per row, items per thread, LDS row pitch
  int row0, chunk;
  typename LD::Ctx ctx[NI];
  uint4 reg[NI]; typename LD::Cur cur;
  __device__ __forceinline__ void init(const LD& l, int base, int tid, int, int kbeg) {
    chunk = tid & (CPR - 1); row0 = tid / CPR; cur = l.seek(kbeg);
#pragma unroll
    for (int i = 0; i < NI; ++i) ctx[i] = l.row(base + row0 + (256 / CPR) * i);
  }
  __device__ __forceinline__ void load(const LD& l) {
#pragma unroll
    for (int i = 0; i < NI; ++i) reg[i] = l.load8(ctx[i], cur, chunk);
#pragma unroll
    for (int j = 0; j < BK / 32; ++j) l.advance(cur);
  }
  __device__ __forceinline__ void skip(const LD& l) {
#pragma unroll
    for (int j = 0; j < BK / 32; ++j) l.advance(cur);
  }
  __device__ __forceinline__ void store(unsigned char* tile) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) *reinterpret_cast<uint4*>(tile + (row0 + (256 / CPR) * i) * PITCH + chunk * 16) = reg[i];
  }
};

// bf16 source, M/N-contiguous: 128 micro-blocks of 8 rows x 4 k per tile, staged by one half of the workgroup
// (half 0: threads 0-127, half 1: threads 128-255) -- 4 dwordx4 loads, 16-bit transpose in registers, 8 x 8-byte writes.
template <class LD> struct Stager<LD, 32, false, true> {
  int row8, k4; bool active;
  typename LD::Ctx8 ctx;
  uint4 reg[4]; typename LD::Cur cur;
  __device__ __forceinline__ void init(const LD& l, int base, int tid, int half, int kbeg) {
    active = (tid >> 7) == half; int t = tid & 127;
    k4 = (t & 7) * 4; row8 = (t >> 3) * 8; ctx = l.row8(base + row8); cur = l.seek(ctx, kbeg + k4);      // k-groups fastest: 2-way instead of 16-way write conflicts
  }
  __device__ __forceinline__ void load(const LD& l) { if (active) { l.load8x4(reg, ctx, cur); l.advance(ctx, cur); } }
  __device__ __forceinline__ void skip(const LD& l) { if (active) l.advance(ctx, cur); }
  __device__ __forceinline__ void store(unsigned char* tile) const {
    if (!active) return;
    const unsigned* w0 = reinterpret_cast<const unsigned*>(&reg[0]);
    const unsigned* w1 = reinterpret_cast<const unsigned*>(&reg[1]);
    const unsigned* w2 = reinterpret_cast<const unsigned*>(&reg[2]);
    const unsigned* w3 = reinterpret_cast<const unsigned*>(&reg[3]);
#pragma unroll
    for (int d = 0; d < 4; ++d) {                       // dword d of each register holds rows 2d (low) and 2d+1 (high)
      uint2 ev, od;
      ev.x = (w0[d] & 0xffffu) | (w1[d] << 16); ev.y = (w2[d] & 0xffffu) | (w3[d] << 16);
      od.x = (w0[d] >> 16) | (w1[d] & 0xffff0000u); od.y = (w2[d] >> 16) | (w3[d] & 0xffff0000u);
      *reinterpret_cast<uint2*>(tile + (row8 + 2 * d) * LDS_PITCH + k4 * 2) = ev;
      *reinterpret_cast<uint2*>(tile + (row8 + 2 * d + 1) * LDS_PITCH + k4 * 2) = od;
    }
  }
};

// one 128 x 128 output tile over k in [kbeg, kend); BK = k per LDS tile (64 only for bf16 K-contiguous sources)
template <int BK, class AL, class BL, class EP>
__device__ __forceinline__ void lds_tile(const AL& a, const BL& b, const EP& ep, int m_blk, int n_blk, int kbeg, int kend,
                                         unsigned char (&lds)[2][2][128 * (BK * 2 + 16)]) {
  constexpr int PITCH = BK * 2 + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + BK - 1) / BK;

  // Register prefetch ring: D staging sets, set d holds tiles d, d+D, ...; the loads of tile kt+D are issued while tile kt
  // is multiplied, so D-1 tiles of global latency stay in flight across the per-tile barrier (bf16 sources only: a
  // set is 8-16 VGPRs there).  LDS stays double-buffered: tile kt+1 is written while tile kt is read.
  // Measured (C3, conv layers): D = 2..3 costs a workgroup per CU (188 vs 152 VGPRs -> 2 instead of 3 resident) and
  // runs 20-50 % SLOWER than D = 1; thread-level parallelism hides the load latency better than a deeper ring here.
  constexpr int D = 1;
  Stager<AL, BK> sa[D]; Stager<BL, BK> sb[D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    sa[d].init(a, m_blk, tid, 0, kbeg); sb[d].init(b, n_blk, tid, 1, kbeg);
#pragma unroll
    for (int j = 0; j < d; ++j) { sa[d].skip(a); sb[d].skip(b); }
  }
  auto gload = [&](auto& xa, auto& xb) {
    xa.load(a); xb.load(b);
#pragma unroll
    for (int j = 0; j < D - 1; ++j) { xa.skip(a); xb.skip(b); }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int d = 0; d < D; ++d) if (d < nk) gload(sa[d], sb[d]);
  if (nk > 0) { sa[0].store(&lds[0][0][0]); sb[0].store(&lds[0][1][0]); }
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kt = kt0 + d;
      if (kt < nk) {
        const int buf = kt & 1;
        if (kt + D < nk) gload(sa[d], sb[d]);              // set d was copied to LDS one tile ago: free again
        const unsigned char* la = &lds[buf][0][(wm * 64 + r) * PITCH + 16 * h];
        const unsigned char* lb = &lds[buf][1][(wn * 64 + r) * PITCH + 16 * h];
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
          bf16x8 af[2], bf[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            af[i] = *reinterpret_cast<const bf16x8*>(la + i * 32 * PITCH + 32 * s);
            bf[i] = *reinterpret_cast<const bf16x8*>(lb + i * 32 * PITCH + 32 * s);
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) { sa[(d + 1) % D].store(&lds[buf ^ 1][0][0]); sb[(d + 1) % D].store(&lds[buf ^ 1][1][0]); }
        __syncthreads();
      }
    }
  }
  const int m0 = m_blk + wm * 64, n0 = n_blk + wn * 64;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}

// ---------------------------------------------------------------------------
// LDS-tiled EXACT-fp32 kernel (fp32 compute mode: BASELINE configs[1], the 1e-4 logit-parity configuration).  Same 128 x 128 x 32
// block tile and staging as lds_tile, the LDS image kept in fp32 (144-byte rows), v_mfma_f32_32x32x2_f32 in the k order of
// gemm_big_kernel<false> (lane (r, h) holds k = 8 c + 4 h + s of chunk c, MFMA s = 0..3), so the results are bit-identical to that
// kernel's.  gemm_big_kernel<false> loads every fragment straight from global memory -- 16 bytes of 32 different rows per wave
// instruction and no reuse between the two waves that share an operand: 0.29 of the fp32 MFMA peak on the conv layers at C2; here a
// tile is fetched once per workgroup with whole 128-byte lines and the fp32 MFMA (64 cycles each) hides the LDS traffic easily.
// ---------------------------------------------------------------------------
template <class AL, class BL, class EP>
__device__ __forceinline__ void lds_tile_f32(const AL& a, const BL& b, const EP& ep, int m_blk, int n_blk, int kbeg, int kend,
                                             unsigned char (&lds)[2][2][128 * LDS_PITCH32]) {
  constexpr int PITCH = LDS_PITCH32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + 31) / 32;
  Stager<AL, 32> sa; Stager<BL, 32> sb;
  sa.init(a, m_blk, tid, 0, kbeg); sb.init(b, n_blk, tid, 1, kbeg);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (nk > 0) { sa.load(a); sb.load(b); sa.store32(&lds[0][0][0]); sb.store32(&lds[0][1][0]); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { sa.load(a); sb.load(b); }            // tile kt+1: global -> registers while tile kt is multiplied
    const unsigned char* la = &lds[buf][0][(wm * 64 + r) * PITCH + 16 * h];
    const unsigned char* lb = &lds[buf][1][(wn * 64 + r) * PITCH + 16 * h];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const float4*>(la + i * 32 * PITCH + 32 * c);
        bf[i] = *reinterpret_cast<const float4*>(lb + i * 32 * PITCH + 32 * c);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const float x = s == 0 ? af[mi].x : s == 1 ? af[mi].y : s == 2 ? af[mi].z : af[mi].w;
            const float y = s == 0 ? bf[ni].x : s == 1 ? bf[ni].y : s == 2 ? bf[ni].z : bf[ni].w;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[mi][ni], 0, 0, 0);
          }
    }
    if (kt + 1 < nk) { sa.store32(&lds[buf ^ 1][0][0]); sb.store32(&lds[buf ^ 1][1][0]); }
    __syncthreads();
  }
  const int m0 = m_blk + wm * 64, n0 = n_blk + wn * 64;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}
template <class AL, class BL, class EP>
__global__ __launch_bounds__(256, 2) void gemm_lds_f32_kernel(AL a, BL b, EP ep, int K, int kper, int gx, int gy) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][128 * LDS_PITCH32];        // 73.7 KB: two workgroups per CU
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int kbeg = blockIdx.z * kper;
  lds_tile_f32(a, b, ep, (bid / gx) * 128, (bid % gx) * 128, kbeg, min(K, kbeg + kper), lds);
}

// ---------------------------------------------------------------------------
// Round 4: gemm_f32t_kernel -- the exact-fp32 LDS-tiled kernel again, for K-contiguous operand pairs whose 8-k chunks are whole (conv
// forward / data gradient with C % 8 == 0, nn.Linear with K % 8 == 0), written around what the ISA of gemm_lds_f32_kernel showed at C2
// (batch 64: 48-400 workgroups of 128 x 128, one wave per SIMD, 0.19-0.42 of the fp32 MFMA peak per layer):
//  * the stager's `if (ok) load else zero` became divergent branches, and the compiler's waitcnt pass put `s_waitcnt vmcnt(0/1)` INSIDE
//    the load phase -- every K tile paid a full global-load latency in front of its 64 MFMAs.  Here a chunk's address is always valid
//    (ptr8: the base pointer when the chunk is padding), all loads are unconditional, and validity is a select at the LDS store, so the
//    only vmcnt wait of a tile sits behind its MFMAs;
//  * the (tap, channel) of the im2col cursor advance by addition (KCur) instead of two integer divisions per load;
//  * the fragments of chunk c+1 are read from LDS while chunk c is multiplied (two register sets);
//  * the tile is a template parameter: 128 x 128, 64 x 128 or 64 x 64 (4 waves as 2 x 2, wave tile BM/2 x BN/2), chosen by the host so
//    that every CU holds 2-4 workgroups: at batch 64 the 3x3 layers run as 400-800 workgroups of 64 x 64 instead of 100-200 of 128 x 128.
// Same LDS image (144-byte rows), same k order per output element (chunk c, MFMA s: k = 8c + s and 8c + 4 + s) as gemm_lds_f32_kernel
// and gemm_big_kernel<false>: bit-identical results (tests/test_step_gpu.py::test_f32t_kernel_matches_lds_f32_kernel).
// ---------------------------------------------------------------------------
template <class LD, int ROWS> struct StagerF32 {
  static constexpr int NI = ROWS / 64;
  typename LD::Ctx ctx[NI]; typename LD::KCur cur; float4 reg[NI][2]; bool ok[NI]; int row0, chunk;
  __device__ __forceinline__ void init(const LD& l, int base, int tid, int kbeg) {
    chunk = tid & 3; row0 = tid >> 2; cur = l.kseek(kbeg + 8 * chunk);
#pragma unroll
    for (int i = 0; i < NI; ++i) ctx[i] = l.row(base + row0 + 64 * i);
  }
  __device__ __forceinline__ void load(const LD& l) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const float* p = l.ptr8(ctx[i], cur, ok[i]);
      reg[i][0] = *reinterpret_cast<const float4*>(p); reg[i][1] = *reinterpret_cast<const float4*>(p + 4);
    }
    l.kadvance(cur, 32);
  }
  __device__ __forceinline__ void store(unsigned char* tile) const {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float4* d = reinterpret_cast<float4*>(tile + (row0 + 64 * i) * LDS_PITCH32 + chunk * 32);
      const float4 u = reg[i][0], w = reg[i][1]; const bool k = ok[i];      // (component selects: a select between the two float4 OBJECTS sends the stager to scratch)
      d[0] = make_float4(k ? u.x : 0.f, k ? u.y : 0.f, k ? u.z : 0.f, k ? u.w : 0.f);
      d[1] = make_float4(k ? w.x : 0.f, k ? w.y : 0.f, k ? w.z : 0.f, k ? w.w : 0.f);
    }
  }
};
template <int BM, int BN, class AL, class BL, class EP>
__global__ __launch_bounds__(256, (BM + BN <= 128) ? 4 : 2) void gemm_f32t_kernel(AL a, BL b, EP ep, int K, int kper, int gx, int gy) {
  constexpr int PITCH = LDS_PITCH32, MI = BM / 64, NI = BN / 64, SLOT = (BM + BN) * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][SLOT];      // 128 x 128: 73.7 KB (2 per CU), 64 x 128: 55.3 KB (2), 64 x 64: 36.9 KB (4)
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);      // consecutive ids share an XCD
  const int m_blk = (bid / gx) * BM, n_blk = (bid % gx) * BN;
  const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + 31) / 32;
  StagerF32<AL, BM> sa; StagerF32<BL, BN> sb;
  sa.init(a, m_blk, tid, kbeg); sb.init(b, n_blk, tid, kbeg);
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (nk > 0) { sa.load(a); sb.load(b); sa.store(&lds[0][0]); sb.store(&lds[0][BM * PITCH]); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { sa.load(a); sb.load(b); }            // tile kt+1: global -> registers; consumed behind this tile's MFMAs
    const unsigned char* la = &lds[buf][(wm * (BM / 2) + r) * PITCH + 16 * h];
    const unsigned char* lb = &lds[buf][(BM + wn * (BN / 2) + r) * PITCH + 16 * h];
    float4 af[2][MI], bf[2][NI];
    auto rd = [&](int set, int c) {
#pragma unroll
      for (int i = 0; i < MI; ++i) af[set][i] = *reinterpret_cast<const float4*>(la + i * 32 * PITCH + 32 * c);
#pragma unroll
      for (int i = 0; i < NI; ++i) bf[set][i] = *reinterpret_cast<const float4*>(lb + i * 32 * PITCH + 32 * c);
    };
    rd(0, 0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < 3) rd((c + 1) & 1, c + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const float4& A = af[c & 1][mi]; const float4& B = bf[c & 1][ni];
            const float x = s == 0 ? A.x : s == 1 ? A.y : s == 2 ? A.z : A.w;
            const float y = s == 0 ? B.x : s == 1 ? B.y : s == 2 ? B.z : B.w;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[mi][ni], 0, 0, 0);
          }
    }
    if (kt + 1 < nk) { sa.store(&lds[buf ^ 1][0]); sb.store(&lds[buf ^ 1][BM * PITCH]); }
    __syncthreads();
  }
  const int m0 = m_blk + wm * (BM / 2), n0 = n_blk + wn * (BN / 2);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[NI][4];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<NI>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}
// ---------------------------------------------------------------------------
// Round 4: gemm_f32w_kernel -- exact-fp32 products of two M/N-CONTIGUOUS operands (conv filter gradients: A = dY^T (LoadMN), B = im2col
// columns (LoadConvXcol); the recurrent weight gradients: LoadMN x LoadMN), K = pixels / time steps.  They ran on gemm_big_kernel<false>
// (every lane loads its own MFMA fragments from global memory, 0.30-0.33 of the fp32 MFMA peak at C2) because the LDS-tiled kernel's
// transposing stagers measured slower.  No transpose is needed for v_mfma_f32_32x32x2_f32: a lane supplies ONE A value (row r, k = h) and
// one B value per MFMA, so the LDS image can stay [k][m] -- exactly how the operands lie in memory.  A tile is 32 k x 128 floats per
// operand: staged with 16-byte loads along m (512 contiguous bytes per k, branch-free: the address of a piece that is padding / halo is
// the base pointer and zeros are selected at the LDS store) and read back as ds_read_b32 (32 consecutive dwords per half-wave; the k-row
// pitch of 136 dwords puts the two halves, 1 k apart, on disjoint banks).  4 waves as 2 x 2, wave tile 64 x 64, double-buffered LDS
// (69.6 KB: two workgroups per CU), split-K over blockIdx.z as before.  k order per output: steps j = 0..15 of a tile, MFMA j takes k = 2j, 2j+1.
// ---------------------------------------------------------------------------
constexpr int F32W_PITCH = 136 * 4;
template <class LD> struct StagerW;
// (addresses advance by ADDITION: a 64-bit k * ld product per load made the compiler guard each address computation with a branch)
template <> struct StagerW<LoadMN> {
  const float* pk; int64_t ld8; bool mok; int k; f32x4 reg[4]; bool ok[4];
  __device__ __forceinline__ void init(const LoadMN& l, int mbase, int tid, int kbeg) {
    const int m = mbase + 4 * (tid & 31); mok = m + 3 < l.rows; k = kbeg + (tid >> 5);
    pk = l.p + (mok ? m : 0) + (int64_t)k * l.ld; ld8 = 8 * l.ld;
  }
  __device__ __forceinline__ void load(const LoadMN& l) {
    const float* q = pk;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ok[i] = mok && k + 8 * i < l.K;
      reg[i] = *reinterpret_cast<const f32x4*>(ok[i] ? q : l.p);
      q += ld8;
    }
    pk = q; k += 32;
  }
};
// im2col columns: element(n = (tap, ci), k = output pixel (b, py, px)) = x[b][py + kh - pad][px + kw - pad][ci].  The cursor keeps the linear
// offset of the un-shifted pixel times Cin (host: the tensor has fewer than 2^31 elements) and moves it by additions: +8 pixels per item,
// + (W - Wo) at the end of an output row, + (H - Ho) W at the end of an image (host: Wo >= 8, so one wrap at most per item).
template <> struct StagerW<LoadConvXcol> {
  LoadConvXcol::Ctx4 c; int px, py, k, linc, cbase, dxc, dyc; f32x4 reg[4]; bool ok[4];
  __device__ __forceinline__ void init(const LoadConvXcol& l, int nbase, int tid, int kbeg) {
    c = l.row4(nbase + 4 * (tid & 31)); k = kbeg + (tid >> 5);
    px = k % l.Wo; const int t = k / l.Wo; py = t % l.Ho; const int b = t / l.Ho;
    linc = ((b * l.H + py) * l.W + px) * l.Cin; cbase = ((c.kh - l.pad) * l.W + (c.kw - l.pad)) * l.Cin + c.ci;
    dxc = (l.W - l.Wo) * l.Cin; dyc = (l.H - l.Ho) * l.W * l.Cin;
  }
  __device__ __forceinline__ void load(const LoadConvXcol& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sy = py + c.kh - l.pad, sx = px + c.kw - l.pad;
      ok[i] = c.ok && k < l.K && (unsigned)sy < (unsigned)l.H && (unsigned)sx < (unsigned)l.W;
      reg[i] = *reinterpret_cast<const f32x4*>(l.x + (ok[i] ? linc + cbase : 0));
      k += 8; px += 8; linc += 8 * l.Cin;
      const bool wx = px >= l.Wo; px -= wx ? l.Wo : 0; py += wx ? 1 : 0; linc += wx ? dxc : 0;
      const bool wy = py >= l.Ho; py = wy ? 0 : py; linc += wy ? dyc : 0;
    }
  }
};
template <class ST> __device__ __forceinline__ void stagerw_store(const ST& st, unsigned char* tile, int tid) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(tile + ((tid >> 5) + 8 * i) * F32W_PITCH + (tid & 31) * 16) = st.ok[i] ? st.reg[i] : z;
}
template <class AL, class BL, class EP>
__global__ __launch_bounds__(256, 2) void gemm_f32w_kernel(AL a, BL b, EP ep, int K, int kper, int gx, int gy) {
  constexpr int PK = F32W_PITCH, OPB = 32 * PK;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2 * OPB];
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * 128, n_blk = (bid % gx) * 128;
  const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + 31) / 32;
  StagerW<AL> sa; StagerW<BL> sb;
  sa.init(a, m_blk, tid, kbeg); sb.init(b, n_blk, tid, kbeg);
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if (nk > 0) { sa.load(a); sb.load(b); stagerw_store(sa, &lds[0][0], tid); stagerw_store(sb, &lds[0][OPB], tid); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { sa.load(a); sb.load(b); }
    const unsigned char* la = &lds[buf][h * PK + (wm * 64 + r) * 4];
    const unsigned char* lb = &lds[buf][OPB + h * PK + (wn * 64 + r) * 4];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) { av[i] = *reinterpret_cast<const float*>(la + 2 * j * PK + i * 128); bv[i] = *reinterpret_cast<const float*>(lb + 2 * j * PK + i * 128); }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi], bv[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) { stagerw_store(sa, &lds[buf ^ 1][0], tid); stagerw_store(sb, &lds[buf ^ 1][OPB], tid); }
    __syncthreads();
  }
  const int m0 = m_blk + wm * 64, n0 = n_blk + wn * 64;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}
template <class L> struct HasStagerW { static constexpr bool v = false; };
template <> struct HasStagerW<LoadMN> { static constexpr bool v = true; };
template <> struct HasStagerW<LoadConvXcol> { static constexpr bool v = true; };

template <class L> struct HasPtr8 { static constexpr bool v = false; };
template <> struct HasPtr8<LoadK> { static constexpr bool v = true; };
template <> struct HasPtr8<LoadConvK> { static constexpr bool v = true; };

template <class AL, class BL, class EP, int BK = 32>
__global__ __launch_bounds__(256, (SrcBf16<AL>::v && SrcBf16<BL>::v) ? 4 : 1)      // bf16-source pairs fit 128 VGPRs: 4 workgroups per CU
void gemm_lds_bf16_kernel(AL a, BL b, EP ep, int K, int kper, int gx, int gy) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][128 * (BK * 2 + 16)];
  // XCD-aware renumbering (bijective form): consecutive renumbered ids share an XCD (ids are dealt round-robin to 8 XCDs)
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int kbeg = blockIdx.z * kper;
  lds_tile<BK>(a, b, ep, (bid / gx) * 128, (bid % gx) * 128, kbeg, min(K, kbeg + kper), lds);
}

// ---------------------------------------------------------------------------
// 256 x 256 bf16 kernel with LDS-DMA staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
// 8 waves as 2(M) x 4(N), 128 x 64 outputs per wave (4 x 2 MFMA 32x32x16 tiles, 128 accumulator registers), one workgroup
// per CU.  K advances in 32-wide tiles through a ring of four 32 KB LDS slots (A 256 x 64 B | B 256 x 64 B): tile t+4 is
// issued while tile t is multiplied, so two to three tiles stay in flight across the single barrier per tile (raw
// s_barrier + counted vmcnt; a __syncthreads() would drain them).  An LDS-DMA instruction writes 1 KiB lane-linearly = 16 rows of the
// image, so the bank swizzle goes on the per-lane SOURCE address: the 16-byte piece at position p of row r holds k-chunk
// p ^ ((r >> 2) & 3), which spreads every 16-lane group of the fragment ds_read_b128 over all 64 banks.
// Every issue() is exactly 4 DMA instructions per lane (tiles past K read the zero page), which keeps the vmcnt
// arithmetic uniform.  Both operands bf16, K-contiguous, K % 32 == 0.
// ---------------------------------------------------------------------------
struct EpConv; struct EpStore;
// Output tile of the 256 x 256 kernels (gemm_halo4_bf16_kernel, gemm_dma_bf16_kernel, conv_wgrad_dma_kernel) through LDS (free once the K loop is over): the direct epilogue costs a lane 128-256 scattered 1-,
// 2- or 4-byte stores (35 us of a 133 us workgroup at conv6 forward: 64-byte segments of 2-byte values, 32-byte segments of arg-max
// bytes); staged, the tile leaves as 16-byte stores of whole rows.  Handles the plain fp32 tile (EpStore without options: the data
// gradients; EpConv pmode 0 with an fp32 destination: conv3 / conv5 in front of their BatchNorm) in two passes of 128 rows, and the
// (2,1)-pooled bf16 + arg-max tile of conv4 / conv6 in one; anything else returns false (-> the quad epilogue).
// the fp32 tile of a 256 x 256 workgroup: two passes of 128 rows (the waves with wm == pass hold them) through a [128][256] fp32 LDS image.
// NI = 32-column accumulator tiles per wave (wave tile 128 x 32 NI), NTH = threads of the workgroup.
template <int NI, int NTH>
__device__ __forceinline__ void tile256_store_f32(float* dst, int64_t ldc, const float* bias, bool relu, const f32x16 (&acc)[4][NI], unsigned char* lds,
                                                  int m_blk, int n_blk, int wm, int wn, int r, int h, int tid, double* part = nullptr, int C = 0, const float* bias2 = nullptr,
                                                  const float* bnb_x = nullptr, const bf16_t* bnb_yb = nullptr, const float* bnb_save = nullptr) {
  // bnb_x (with part): the sums are those of the BatchNorm BACKWARD pass -- (sum d, sum d xhat), d = stored value where the bf16 activation bnb_yb is positive,
  // xhat = (bnb_x - mean) invstd (bnb_save = {mean[C], invstd[C]}): EpStore::bnb_*
  constexpr int PITCH = 1024;                             // lanes r = consecutive dwords, the two row groups h are separate LDS cycles: no padding needed
  float bmean[4] = {0.f, 0.f, 0.f, 0.f}, binv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bnb_x) {
    const float4 m4 = *reinterpret_cast<const float4*>(bnb_save + n_blk + (tid & 63) * 4), i4 = *reinterpret_cast<const float4*>(bnb_save + C + n_blk + (tid & 63) * 4);
    bmean[0] = m4.x; bmean[1] = m4.y; bmean[2] = m4.z; bmean[3] = m4.w; binv[0] = i4.x; binv[1] = i4.y; binv[2] = i4.z; binv[3] = i4.w;
  }
  float bb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[ni] = (bias ? bias[n_blk + wn * 32 * NI + ni * 32 + r] : 0.f) + (bias2 ? bias2[n_blk + wn * 32 * NI + ni * 32 + r] : 0.f);
  float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                        // every wave is out of the K loop
  for (int p = 0; p < 2; ++p) {
    if (wm == p) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float x = acc[mi][ni][4 * q + i] + bb[ni];
              if (relu) x = fmaxf(x, 0.f);
              *reinterpret_cast<float*>(lds + (mi * 32 + 8 * q + 4 * h + i) * PITCH + (wn * 32 * NI + ni * 32 + r) * 4) = x;
            }
    }
    __syncthreads();
    float* const d0 = dst + (int64_t)(m_blk + p * 128) * ldc + n_blk;
#pragma unroll 4
    for (int it = 0; it < 8192 / NTH; ++it) {
      const int idx = it * NTH + tid, row = idx >> 6, c = idx & 63;
      const float4 v = *reinterpret_cast<const float4*>(lds + row * PITCH + c * 16);
      *reinterpret_cast<float4*>(d0 + (int64_t)row * ldc + c * 4) = v;
      if (part && bnb_x) {                                // BatchNorm backward sums of the thread's four columns
        const int64_t gofs = (int64_t)(m_blk + p * 128 + row) * C + n_blk + c * 4;
        const float4 x4 = *reinterpret_cast<const float4*>(bnb_x + gofs);
        typedef __bf16 bf16x4_ __attribute__((ext_vector_type(4)));
        const bf16x4_ y4 = *reinterpret_cast<const bf16x4_*>(bnb_yb + gofs);
        const float dv[4] = {(float)y4[0] > 0.f ? v.x : 0.f, (float)y4[1] > 0.f ? v.y : 0.f, (float)y4[2] > 0.f ? v.z : 0.f, (float)y4[3] > 0.f ? v.w : 0.f};
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { ps[k] += dv[k]; ps[4 + k] = fmaf(dv[k], (xs[k] - bmean[k]) * binv[k], ps[4 + k]); }
      } else if (part) {                                  // a thread keeps its four columns (c = tid & 63): column sums of exactly the stored values
        ps[0] += v.x; ps[1] += v.y; ps[2] += v.z; ps[3] += v.w;
        ps[4] = fmaf(v.x, v.x, ps[4]); ps[5] = fmaf(v.y, v.y, ps[5]); ps[6] = fmaf(v.z, v.z, ps[6]); ps[7] = fmaf(v.w, v.w, ps[7]);
      }
    }
    __syncthreads();
  }
  if (part) {                                             // 32 (NTH = 512) / 64 rows per thread so far: the NTH / 64 row groups meet in LDS, fp64 from there on
    float* const red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int k = 0; k < 8; ++k) red[((tid >> 6) * 64 + (tid & 63)) * 8 + k] = ps[k];
    __syncthreads();
    if (tid < 64) {
      double S[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int g = 0; g < NTH / 64; ++g)
#pragma unroll
        for (int k = 0; k < 8; ++k) S[k] += (double)red[(g * 64 + tid) * 8 + k];
      double* o = part + ((int64_t)(m_blk >> 8) * C + n_blk + tid * 4) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[2 * j] = S[j]; o[2 * j + 1] = S[4 + j]; }
    }
  }
}
// the (2,1)-pooled tile: 128 pooled rows of 256 bf16 + 256 arg-max bytes
template <int NI, int NTH>
__device__ __forceinline__ void tile256_store_pooled(bf16_t* yb, uint8_t* idxp, int Cout, const float* bias, bool relu, const f32x16 (&acc)[4][NI], unsigned char* lds,
                                                     int m_blk, int n_blk, int wm, int wn, int r, int h, int tid) {
  constexpr int PB = 528, PI = 272, IOFF = 128 * PB;      // pooled rows: 256 bf16 + 16 bytes; 256 arg-max bytes + 16
  float bb[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) bb[ni] = bias ? bias[n_blk + wn * 32 * NI + ni * 32 + r] : 0.f;
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = acc[mi][ni][4 * q + i] + bb[ni]; if (relu) x[i] = fmaxf(x[i], 0.f); }
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const float a_ = x[2 * w], b_ = x[2 * w + 1];
          const int prow = wm * 64 + mi * 16 + 4 * q + 2 * h + w, col = wn * 32 * NI + ni * 32 + r;
          *reinterpret_cast<bf16_t*>(lds + prow * PB + col * 2) = (bf16_t)((b_ > a_) ? b_ : a_);
          *reinterpret_cast<uint8_t*>(lds + IOFF + prow * PI + col) = (uint8_t)(b_ > a_);
        }
      }
  __syncthreads();
  const int64_t prow0 = m_blk >> 1;
#pragma unroll 4
  for (int it = 0; it < 4096 / NTH; ++it) {
    const int idx = it * NTH + tid, row = idx >> 5, c = idx & 31;
    const uint4 v = *reinterpret_cast<const uint4*>(lds + row * PB + c * 16);
    *reinterpret_cast<uint4*>(yb + (prow0 + row) * Cout + n_blk + c * 8) = v;
  }
#pragma unroll 4
  for (int it = 0; it < 2048 / NTH; ++it) {
    const int idx = it * NTH + tid, row = idx >> 4, c = idx & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(lds + IOFF + row * PI + c * 16);
    *reinterpret_cast<uint4*>(idxp + (prow0 + row) * Cout + n_blk + c * 16) = v;
  }
}
// the bf16-only tile (evaluation mode: conv3 / conv5 with their BatchNorm + ReLU folded in, EpConv::bn_save): [256][256] bf16 = 128 KB, one pass
template <int NI, int NTH>
__device__ __forceinline__ void tile256_store_bf16(bf16_t* yb, int Cout, const float* bias, bool relu, const float* bn_save, const float* bn_w, const float* bn_b,
                                                   const f32x16 (&acc)[4][NI], unsigned char* lds, int m_blk, int n_blk, int wm, int wn, int r, int h, int tid, double* part = nullptr) {
  float bb[NI], mu[NI], iv[NI], ww[NI], b2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int col = n_blk + wn * 32 * NI + ni * 32 + r;
    bb[ni] = bias ? bias[col] : 0.f;
    mu[ni] = bn_save ? bn_save[col] : 0.f; iv[ni] = bn_save ? bn_save[Cout + col] : 1.f; ww[ni] = bn_save ? bn_w[col] : 1.f; b2[ni] = bn_save ? bn_b[col] : 0.f;
  }
  float ps[NI], pss[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) { ps[ni] = 0.f; pss[ni] = 0.f; }
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = acc[mi][ni][4 * q + i] + bb[ni];
          if (relu) x = fmaxf(x, 0.f);
          if (bn_save) x = fmaxf((x - mu[ni]) * iv[ni] * ww[ni] + b2[ni], 0.f);      // the same expression as EpConv::quad / bn_apply_relu_kernel
          ps[ni] += x; pss[ni] = fmaf(x, x, pss[ni]);                                 // (part: column sums of the UNROUNDED values, 64 per lane and column)
          *reinterpret_cast<bf16_t*>(lds + (wm * 128 + mi * 32 + 8 * q + 4 * h + i) * 512 + (wn * 32 * NI + ni * 32 + r) * 2) = (bf16_t)x;
        }
  __syncthreads();
#pragma unroll 4
  for (int it = 0; it < 8192 / NTH; ++it) {
    const int idx = it * NTH + tid, row = idx >> 5, c = idx & 31;
    const uint4 v = *reinterpret_cast<const uint4*>(lds + row * 512 + c * 16);
    *reinterpret_cast<uint4*>(yb + (int64_t)(m_blk + row) * Cout + n_blk + c * 8) = v;
  }
  if (part) {                                             // the four (row half wm, row group h) partials of a column meet in LDS, fp64 from there on
    __syncthreads();
    float* const red = reinterpret_cast<float*>(lds);     // [wm][h][256 columns][2]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float* o = red + (((wm * 2 + h) * 256) + wn * 32 * NI + ni * 32 + r) * 2;
      o[0] = ps[ni]; o[1] = pss[ni];
    }
    __syncthreads();
    if (tid < 256) {
      double S = 0.0, SS = 0.0;
#pragma unroll
      for (int g = 0; g < 4; ++g) { S += (double)red[((g * 256) + tid) * 2]; SS += (double)red[((g * 256) + tid) * 2 + 1]; }
      double* o = part + ((int64_t)(m_blk >> 8) * Cout + n_blk + tid) * 2;
      o[0] = S; o[1] = SS;
    }
  }
}
// full 256 x 256 tiles only (the callers' grids may end in a ragged tile: that one takes the quad epilogue)
template <int NI, int NTH, class EP>
__device__ __forceinline__ bool tile256_store_staged(const EP& ep, const f32x16 (&acc)[4][NI], unsigned char* lds, int m_blk, int n_blk, int wm, int wn, int r, int h, int tid, int opt) {
  if constexpr (std::is_same<EP, EpConv>::value) {
    if (m_blk + 256 > ep.rows || n_blk + 256 > ep.Cout) return false;
    if (ep.pmode == 0 && ep.y16 && ep.bn_part && !ep.yb && !ep.bn_save && (opt & 2)) {     // pre-BatchNorm output as bf16 + its statistics (training)
      tile256_store_bf16<NI, NTH>(ep.y16, ep.Cout, ep.bias, ep.relu != 0, nullptr, nullptr, nullptr, acc, lds, m_blk, n_blk, wm, wn, r, h, tid, ep.bn_part); return true; }
    if (ep.pmode == 0 && ep.yb && !ep.y && (opt & 2)) { tile256_store_bf16<NI, NTH>(ep.yb, ep.Cout, ep.bias, ep.relu != 0, ep.bn_save, ep.bn_w, ep.bn_b, acc, lds, m_blk, n_blk, wm, wn, r, h, tid); return true; }
    if (ep.bn_save) return false;
    if (ep.pmode == 2 && ep.yb && ep.idx && !ep.y && (opt & 4)) { tile256_store_pooled<NI, NTH>(ep.yb, ep.idx, ep.Cout, ep.bias, ep.relu != 0, acc, lds, m_blk, n_blk, wm, wn, r, h, tid); return true; }
    if (ep.pmode == 0 && ep.y && !ep.yb && (opt & 2)) { tile256_store_f32<NI, NTH>(ep.y, ep.Cout, ep.bias, ep.relu != 0, acc, lds, m_blk, n_blk, wm, wn, r, h, tid, ep.bn_part, ep.Cout); return true; }
    return false;
  } else if constexpr (std::is_same<EP, EpStore>::value) {
    if (!ep.C && ep.Cb && !ep.flags && !ep.bias && !ep.bias2 && !ep.C1 && !ep.dg) {       // bf16-only data gradient (conv_backward_data's dx16: full tiles guaranteed by the caller)
      tile256_store_bf16<NI, NTH>(ep.Cb, (int)ep.ldcb, nullptr, false, nullptr, nullptr, nullptr, acc, lds, m_blk, n_blk, wm, wn, r, h, tid); return true; }
    if (ep.flags || ep.bias || ep.bias2 || ep.C1 || ep.Cb || ep.dg || !(opt & 1) || m_blk + 256 > ep.M || n_blk + 256 > ep.N) return false;
    if (ep.bnb_part) tile256_store_f32<NI, NTH>(ep.C, ep.ldc, nullptr, false, acc, lds, m_blk, n_blk, wm, wn, r, h, tid, ep.bnb_part, ep.N, nullptr, ep.bnb_x, ep.bnb_yb, ep.bnb_save);
    else tile256_store_f32<NI, NTH>(ep.C, ep.ldc, nullptr, false, acc, lds, m_blk, n_blk, wm, wn, r, h, tid);      // (taking the biases here too cost the data-gradient kernel 157 -> 190 us: register pressure in its K loop)
    return true;
  } else return false;
}

// (defined in epilogues.h, behind EpStore: the staged fp32 tile of gemm_dma_narrow_kernel)
template <int NT, class EPT>
__device__ __forceinline__ bool narrow_store_staged(const EPT& ep, const f32x16 (&acc)[2][NT], unsigned char* lds, int m_blk, int n_blk, int wm, int wn, int r, int h, int tid);

__device__ __forceinline__ void dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <class AL, class BL, class EP, int ABL = 0, bool PIPE = false, bool PAIRS = false, int TAG = 0>     // TAG: distinct symbol for aocr_profile_kernel's launches (so that rocprofv3 --stats lists them on their own row); PAIRS: two K tiles per barrier; PIPE: fragment reads pipelined across the barrier (inline asm; measured 4-7 % slower); ABL: timing-only ablations (tools/ubench/dma_gemm.hip): 1 no in-loop DMA, 2 no MFMA, 4 no fragment reads, 16 / 128 A / B pieces from the zero page (plain form), 32 no A pieces at all (plain form)
__global__ __launch_bounds__(512, 1)
void gemm_dma_bf16_kernel(AL a, BL b, EP ep, int K, int gx, int gy, const bf16_t* zero, int opt = 0) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 32768];          // the ONLY LDS object (a second one makes hipcc drain vmcnt)
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * 256, n_blk = (bid % gx) * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;

  // staging role: rows (tid >> 2) + 128 j, LDS position tid & 3 of the row, logical chunk = position ^ swizzle(row)
  const int srow = tid >> 2, chunk = (tid & 3) ^ ((tid >> 4) & 3);
  typename AL::DRow ra[2]; typename BL::DRow rb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { ra[j] = a.drow(m_blk + 128 * j + srow, chunk); rb[j] = b.drow(n_blk + 128 * j + srow, chunk); }
  typename AL::DCur ca = a.dseek(0); typename BL::DCur cb = b.dseek(0);
  const int nk = K >> 5;
  unsigned char* const wbase = lds + __builtin_amdgcn_readfirstlane(wave) * 1024;
  int slot = 0;                                       // ring slot the next issue() fills
  auto issue_a = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) if constexpr (!(ABL & 32)) dma16((ABL & 16) ? zero : a.dsrc(ra[j], ca, zero), wbase + slot * 32768 + j * 8192);
    a.dadvance(ca);
  };
  auto issue_b = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16((ABL & 128) ? zero : b.dsrc(rb[j], cb, zero), wbase + slot * 32768 + 16384 + j * 8192);
    b.dadvance(cb); slot = (slot + 1) & 3;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int swz = (r >> 2) & 3;
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    aoff[s] = (wm * 128 + r) * 64 + (((2 * s + h) ^ swz) << 4);
    boff[s] = 16384 + (wn * 64 + r) * 64 + (((2 * s + h) ^ swz) << 4);
  }

  // Fragment reads are software-pipelined across the barrier: the k 16..31 half of tile t is fetched under the MFMAs of
  // its k 0..15 half, and the first half of tile t+1 under the MFMAs of the second half of tile t -- right after the
  // barrier the MFMA pipe already has operands in registers.  The reads are inline asm with hand-counted lgkmcnt waits
  // (hipcc's own insertion waits lgkmcnt(0) for the loop-carried half, i.e. for the reads just issued); every wait is
  // followed by sched_barrier(0) so that no MFMA is hoisted above it.
  const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
  bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
#define AOCR_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
  auto read0 = [&](int sl) {
    const unsigned pa = lbase + sl * 32768 + aoff[0], pb = lbase + sl * 32768 + boff[0];
    AOCR_DSR(fa0[0], pa, 0); AOCR_DSR(fa0[1], pa, 2048); AOCR_DSR(fb0[0], pb, 0); AOCR_DSR(fb0[1], pb, 2048);
    AOCR_DSR(fa0[2], pa, 4096); AOCR_DSR(fa0[3], pa, 6144);
  };
  auto read1 = [&](int sl) {
    const unsigned pa = lbase + sl * 32768 + aoff[1], pb = lbase + sl * 32768 + boff[1];
    AOCR_DSR(fa1[0], pa, 0); AOCR_DSR(fa1[1], pa, 2048); AOCR_DSR(fb1[0], pb, 0); AOCR_DSR(fb1[1], pb, 2048);
    AOCR_DSR(fa1[2], pa, 4096); AOCR_DSR(fa1[3], pa, 6144);
  };
#undef AOCR_DSR
  if constexpr (!PIPE && PAIRS) {                       // two K tiles per barrier: 72 instead of 144 barriers at K = 4608 (a bare
    // s_waitcnt + s_barrier iteration costs ~0.1 us: 30 us of the 300 us launch, tools/ubench/dma_gemm.hip).  Ring = two pairs of
    // slots; pair p+1 is issued right after the barrier that ends pair p-1 and lands under the 32 MFMAs per wave of pair p.
    const int npairs = (nk + 1) >> 1;                   // an odd last tile is padded by a zero-page tile
    issue_a(); issue_b(); issue_a(); issue_b();
    for (int pr = 0; pr < npairs; ++pr) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of pair pr have landed
      __builtin_amdgcn_s_barrier();                     // ... everyone's have, and everyone is done reading pair pr-1
      if constexpr (!(ABL & 1)) { issue_a(); issue_b(); issue_a(); issue_b(); }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const unsigned char* L = lds + (((pr & 1) << 1) + half) * 32768;
        bf16x8 af[2][4], bf[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) af[s2][mi] = *reinterpret_cast<const bf16x8*>(L + aoff[s2] + mi * 2048);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[s2][ni] = *reinterpret_cast<const bf16x8*>(L + boff[s2] + ni * 2048);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][mi], bf[s2][ni], acc[mi][ni], 0, 0, 0);
      }
    }
  } else if constexpr (!PIPE) {                         // plain form: compiler-managed reads, tile kt+3 issued under tile kt
#pragma unroll
    for (int t = 0; t < 3; ++t) { issue_a(); issue_b(); }
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (ABL & 32) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* L = lds + (kt & 3) * 32768;
      bf16x8 af[2][4], bf[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[s2][mi] = *reinterpret_cast<const bf16x8*>(L + aoff[s2] + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[s2][ni] = *reinterpret_cast<const bf16x8*>(L + boff[s2] + ni * 2048);
      }
      if constexpr (!(ABL & 1)) issue_a();
      if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(1);      // experiment: raise the wave's priority over its MFMA block
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], bf[0][ni], acc[mi][ni], 0, 0, 0);
      if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(0);
      if constexpr (!(ABL & 1)) issue_b();
      if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][mi], bf[1][ni], acc[mi][ni], 0, 0, 0);
      if constexpr (ABL & 8) __builtin_amdgcn_s_setprio(0);
    }
  } else {
#pragma unroll
  for (int t = 0; t < 3; ++t) { issue_a(); issue_b(); }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();                         // tile 0 is resident
  issue_a(); issue_b();                                 // tile 3
  read0(0);
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (!(ABL & 4)) read1(kt & 3);
    asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");  // the six older reads (first half, issued one barrier ago) are back
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 2)) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[mi], fb0[ni], acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");   // own pieces of tile kt+1 landed (kt+2, kt+3 in flight); own reads of tile kt done
    __builtin_amdgcn_s_barrier();                       // ... everyone's: tile kt+1 is resident and slot kt is free
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 1)) issue_a();                // tile kt+4 -> the slot of tile kt
    if constexpr (!(ABL & 4)) read0((kt + 1) & 3);
    if constexpr (!(ABL & 1)) issue_b();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(ABL & 2)) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[mi], fb1[ni], acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the trailing (zero-page) tiles must land before the LDS is released
  const int m0 = m_blk + wm * 128, n0 = n_blk + wn * 64;
  if constexpr (ABL == 0) { if (opt && tile256_store_staged<2, 512>(ep, acc, lds, m_blk, n_blk, wm, wn, r, h, tid, opt)) return; }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}

// ---------------------------------------------------------------------------
// 3 x 3 convolution (forward / data gradient) with the input HALO resident in LDS.  Ablation of gemm_dma_bf16_kernel at the
// conv6 shape (tools/ubench/dma_gemm.hip): 295 us full, 206-223 with the im2col pieces read from one cached line, 193 with
// both operands so -- a third of the launch is the im2col stream, which fetches every input pixel nine times (once per tap)
// as 64-byte segments of 16 different rows per DMA instruction.  Here a workgroup's 256 output pixels are R = 256 / W whole
// rows of ONE image, and for a 32-channel chunk their (R + 2) x (W + 2) input halo is staged ONCE (two buffers: chunk c + 1
// streams in while the nine taps of chunk c are multiplied); a tap is a shift of the fragment read address.  K runs chunk-major
// (chunk, kh, kw) instead of tap-major, so the weight tile of step (c, tap) starts at k = tap * C + 32 c.  L2 -> LDS bytes per
// 9 steps: 30 + 144 KB instead of 144 + 144 KB.
//   halo image: pitch P = W + 16 pixels (16-pixel DMA groups never straddle a row), pixel (row, col) at (row * P + col) * 64 B,
//   col = x + 1, row = y - y0 + 1; the 16-byte piece at position p of a pixel holds k-chunk p ^ s, s = ((col >> 2) & 3) ^ 2 (row & 1):
//   any 16 consecutive fragment rows (16 pixels of a row, or 8 + 8 pixels of two adjacent rows under the pooled row orders) hit
//   all 64 banks once; a tap's dy only toggles bit 1 of s (address ^ 32), its dx picks one of three precomputed lane addresses.
//   Every step issues exactly 3 DMA instructions per wave (1 halo piece -- a dummy into a dump page once the next halo is
//   complete -- and 2 weight pieces), so the counted wait is uniform: vmcnt(6) = the pieces of this step's weight tile landed.
// Shapes: W in {32, 64, 128}, H % R == 0 (R even under the pooled orders), C % 32 == 0, N % NT == 0; SGN = +1 forward, -1 data gradient.
// ---------------------------------------------------------------------------
// Tile shapes MT x NT (8 waves, every wave 32 MI x 64 outputs): 256 x 256 (2 x 4 waves, MI = 4) for layers with >= 256 output columns;
// 512 x 128 (4 x 2, MI = 4) and 512 x 64 (8 x 1, MI = 2) for the narrow layers (conv2 forward, the data gradients of conv2 / conv3),
// whose im2col stream is 4-9x their weight stream: there the halo cuts the L2 -> LDS bytes per step from 32 + 8 KB to 6 + 8 KB.
// debugging probe of the TAG = 1 instantiations (aocr_profile_kernel under AOCR_PROBE=1): shader cycles / 100 MHz wall ticks of one workgroup's K loop and of the whole workgroup
static __device__ unsigned long long g_kprobe[8];
template <int MT> struct HaloGeom { static constexpr int HMAX = MT == 256 ? 36864 : 55296; };   // one halo buffer: (R + 2) (W + 16) 64 B for W in {32, 64, 128}

// (Round 3, measured and removed: all LDS-DMA pieces of a step issued by ONE wave per SIMD -- waves 0-3, six pieces each -- so that the
// SIMD partner multiplies while the loader wave sits in DMA issue: conv6 forward 0.246 -> 0.29 ms per launch, conv forward + data
// gradient +10 % per step.  The issue cost is serial per wave; spread over eight waves it is half as long.)
// (Round 3, measured and removed: the weight tile of a step -- 16 KB, 94 % of this kernel's LDS-DMA bytes -- staged global -> registers ->
// ds_write_b128 instead of by LDS-DMA, the halo piece still a DMA, bit-identical results.  One tile in flight: conv forward 0.84 -> 0.88 ms
// per step; THREE tiles in flight (asm loads, counted vmcnt(7), 244 VGPRs, no spill): conv6 forward 0.247 -> 0.265 ms per launch, conv
// forward 0.83 -> 0.865 ms.  Neither fewer DMA instructions per wave, nor fewer waves issuing them, nor taking 94 % of the bytes off the
// LDS-DMA path helps: within a wave fragment reads, DMA issue and MFMA issue serialise (0.19 + 0.30 + 0.45 us of a 1.0 us step, ablations of
// round 2), and the two waves of a SIMD cannot fill each other's gaps because the per-step barrier keeps them in the same phase.)
template <class EP, int SGN, int MT, int NT, int TAG = 0>          // TAG: distinct symbol for aocr_profile_kernel's launches (their own row in rocprofv3 --stats)
__global__ __launch_bounds__(512, 1)
void gemm_halo_bf16_kernel(LoadConvKh a, LoadKh b, EP ep, int gx, int gy, const bf16_t* zero) {
  unsigned long long pe0 = 0; if constexpr (TAG == 1) pe0 = wall_clock64();
  constexpr int HMAX = HaloGeom<MT>::HMAX, BSLOT = NT * 64, BRING = 2 * HMAX, DUMP = BRING + 4 * BSLOT, LDS_BYTES = DUMP + 8 * 1024;
  constexpr int NWN = NT / 64, NWM = 8 / NWN, WM = MT / NWM, MI = WM / 32;          // wave grid and wave tile (WM x 64)
  constexpr int NBW = NT >= 128 ? NT / 128 : 1;          // weight pieces per wave and step (NT = 64: waves 4-7 issue a dummy)
  static_assert(MI == 4 || MI == 2, "wave tile");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];          // the ONLY LDS object (a second one makes hipcc drain vmcnt)
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * MT, n_blk = (bid % gx) * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = wave / NWN, wn = wave % NWN;
  const LoadConvK& g = a.g;
  const int W = g.Wr, H = g.Hr, C = g.C, P = W + 16, R = MT / W;
  const int NG = (R + 2) * (P >> 4);                     // 16-pixel DMA groups of one halo
  const int NC = C >> 5, NT9 = 9 * NC;                   // channel chunks, K steps
  const LoadConvK::Ctx c0 = g.row(m_blk);                // first pixel of the tile: (b, y0, 0)
  const int y0 = __builtin_amdgcn_readfirstlane(c0.y);
  const bf16_t* const img = a.src + (int64_t)__builtin_amdgcn_readfirstlane(c0.b) * H * W * C;

  // ---- fragment addresses: abase[mi][dxi] = byte offset of (halo row ty, col tx + dxi) for k-chunk h, i.e. tap dy = -1, dx = dxi - 1
  unsigned abase[MI][3];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int local = wm * WM + mi * 32 + r;
    int ty, tx;
    if (g.pmode == 1) { const int win = local >> 2, wx = win % g.Wp; ty = 2 * (win / g.Wp) + ((local >> 1) & 1); tx = 2 * wx + (local & 1); }
    else if (g.pmode == 2) { const int win = local >> 1; tx = win % W; ty = 2 * (win / W) + (local & 1); }
    else { ty = local / W; tx = local - ty * W; }
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int col = tx + dxi, sw = ((col >> 2) & 3) ^ ((ty & 1) << 1);
      abase[mi][dxi] = (unsigned)(ty * P + col) * 64u + (unsigned)((h ^ sw) << 4);
    }
  }
  const int swzb = (r >> 2) & 3;
  unsigned boff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) boff[s2] = BRING + (wn * 64 + r) * 64 + (((2 * s2 + h) ^ swzb) << 4);

  // ---- weight staging (as in gemm_dma_bf16_kernel): rows (tid >> 2) + 128 j, position tid & 3 holds k-chunk (tid & 3) ^ ((tid >> 4) & 3)
  const int srow = tid >> 2, bchunk = (tid & 3) ^ ((tid >> 4) & 3);
  const bool bwave = NT >= 128 || wave < 4;              // NT = 64: rows 0..63 are waves 0-3
  LoadKh::DRow rb[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) rb[j] = b.drow(n_blk + 128 * j + srow, bchunk);
  unsigned char* const wbase = lds + wave * 1024;
  // ---- halo staging: group gq = 16 pixels of one halo row; lane -> pixel (lane >> 2), position lane & 3
  const int hx = (lane >> 2) - 1, hchunk = (lane & 3) ^ ((lane >> 4) & 3);
  auto issue_halo = [&](int gq, int chunk, bool real) {   // gq, chunk, real: wave-uniform
    const int row = gq / (P >> 4), col0 = (gq - row * (P >> 4)) << 4;
    const int y = y0 - 1 + row, x = col0 + hx;
    const bool ok = real && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    const bf16_t* src = img + ((int64_t)(y * W + x) * C + (chunk << 5) + ((hchunk ^ ((row & 1) << 1)) << 3));
    dma16(dma_select(ok, src, zero), real ? lds + (chunk & 1) * HMAX + gq * 1024 : lds + DUMP + wave * 1024);
  };
  auto issue_b = [&](int step) {                         // weight tile of K step `step` (chunk-major): k = tap * C + 32 chunk
    const int chunk = step / 9, tap = step - chunk * 9;
    const int k = step < NT9 ? tap * C + (chunk << 5) : b.K;           // past the end: zero page
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      dma16(dma_select(bwave && rb[j].b != nullptr && k < b.K, rb[j].b + k, zero),
            bwave ? wbase + BRING + (step & 3) * BSLOT + j * 8192 : lds + DUMP + wave * 1024);
  };

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // prologue: the whole halo of chunk 0 and three weight tiles, all landed before the first step
  for (int gq = wave; gq < NG; gq += 8) issue_halo(gq, 0, true);
  issue_b(0); issue_b(1); issue_b(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  unsigned long long pc0 = 0, pw0 = 0; if constexpr (TAG == 1) { pc0 = __builtin_readcyclecounter(); pw0 = wall_clock64(); }
  int step = 0;
  for (int chunk = 0; chunk < NC; ++chunk) {
    const unsigned hb = (chunk & 1) * HMAX;
    for (int kh = 0; kh < 3; ++kh) {
      const int dyi = SGN > 0 ? kh : 2 - kh;             // halo row offset of this tap: dy + 1
      const unsigned U = hb + (unsigned)(dyi * P) * 64u, flip = (dyi & 1) << 5;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw, ++step) {
        const int dxi = SGN > 0 ? kw : 2 - kw;
        // this wave's pieces of this step's weight tile (and of everything older) have landed: 2 steps x (1 + NBW) pieces may be pending
        if constexpr (NBW == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // ... everyone's have, and everyone is done reading the previous step
        const unsigned char* Lb = lds + (step & 3) * BSLOT;
        bf16x8 af[2][MI], bf[2][2];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const unsigned a0 = (abase[mi][dxi] + U) ^ flip;
          af[0][mi] = *reinterpret_cast<const bf16x8*>(lds + a0);
          af[1][mi] = *reinterpret_cast<const bf16x8*>(lds + (a0 ^ 32u));
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) bf[s2][ni] = *reinterpret_cast<const bf16x8*>(Lb + boff[s2] + ni * 2048);
        {                                                 // one piece of the NEXT chunk's halo (its buffer was last read a chunk ago)
          const int gq = (kh * 3 + kw) * 8 + wave;
          issue_halo(gq < NG ? gq : 0, chunk + 1, gq < NG && chunk + 1 < NC);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][mi], bf[0][ni], acc[mi][ni], 0, 0, 0);
        issue_b(step + 3);                                // -> the slot of the previous step
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][mi], bf[1][ni], acc[mi][ni], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the trailing (zero-page) pieces must land before the LDS is released
  if constexpr (TAG == 1) { if (blockIdx.x == 17 && tid == 0) { g_kprobe[0] = __builtin_readcyclecounter() - pc0; g_kprobe[1] = wall_clock64() - pw0; g_kprobe[2] = pw0 - pe0; } }
  const int m0 = m_blk + wm * WM, n0 = n_blk + wn * 64;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
  if constexpr (TAG == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (blockIdx.x == 17 && tid == 0) g_kprobe[3] = wall_clock64() - pe0; }
}

// ---------------------------------------------------------------------------
// The halo-resident kernel as FOUR waves of 128 x 128 outputs (16 accumulator tiles = 256 AGPRs, one wave per SIMD), 256 x 256 tiles
// only.  tools/ubench/gemm4w.hip: the 8-wave form reads 96 KB of fragments per K step (A rows by four waves, B rows by two) and its
// two waves per SIMD sit in the same phase behind the per-step barrier; four waves read 64 KB and -- with every fragment read and
// LDS-DMA piece pinned BETWEEN two MFMAs (one read per two MFMAs, asm reads + sched_barrier, fragments of the next k-half always in
// flight under the MFMAs of the current one) -- the reads cost no MFMA cycles at all (155 k vs 152 k shader cycles per workgroup for
// the MFMA-only loop); what remains is ~24 cycles per global_load_lds.
//   step s = (chunk, tap); halves h0 / h1 = k 0..15 / 16..31 of the 32-channel chunk.
//     half 1:  MFMAs on F0(s) | reads of F1(s) | DMA: weight pieces 0, 1 of step s+3 [+ one halo piece of chunk+1 while tap <= 4]
//     wait (own weight pieces of step s+1 landed, F1 reads back) + barrier
//     half 2:  MFMAs on F1(s) | reads of F0(s+1) | DMA: weight pieces 2, 3 of step s+3 [+ one halo piece while tap <= 4]
//   The barrier in the MIDDLE of a step is what lets F0(s+1) be read under the second half's MFMAs.  Weight slot (s+3) & 3 was last read
//   in half 1 of step s-1; the halo buffer of chunk+1 was last read in half 1 of the previous chunk's last step.  The nine taps are
//   unrolled (compile-time tap -> compile-time vmcnt: the pieces issued since the last piece of step s+1's weights number
//   6 + H(t-2) + 2 H(t-1) + H(t), H = 1 while tap <= 4); the <= 36 halo groups of a chunk are all issued by tap 4, so they have
//   landed long before the last step's barrier.  Same k order per output element as the 8-wave kernel: bit-identical results.
// ---------------------------------------------------------------------------
template <class EP, int SGN, int TAG = 0>
__global__ __launch_bounds__(256, 1)
void gemm_halo4_bf16_kernel(LoadConvKh a, LoadKh b, EP ep, int gx, int gy, const bf16_t* zero, int opt) {
  unsigned long long pe0 = 0; if constexpr (TAG == 1) pe0 = wall_clock64();
  constexpr int MT = 256, NT = 256, HMAX = HaloGeom<MT>::HMAX, BSLOT = NT * 64, BRING = 2 * HMAX, DUMP = BRING + 4 * BSLOT, LDS_BYTES = DUMP + 4 * 1024;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];          // the ONLY LDS object
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * MT, n_blk = (bid % gx) * NT;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const LoadConvK& g = a.g;
  const int W = g.Wr, H = g.Hr, C = g.C, P = W + 16, R = MT / W;
  const int NG = (R + 2) * (P >> 4);                     // 16-pixel DMA groups of one halo (<= 36)
  const int NC = C >> 5, NT9 = 9 * NC;
  const LoadConvK::Ctx c0 = g.row(m_blk);
  const int y0 = __builtin_amdgcn_readfirstlane(c0.y);
  const bf16_t* const img = a.src + (int64_t)__builtin_amdgcn_readfirstlane(c0.b) * H * W * C;
  const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);

  unsigned abase[4][3];                                  // as in the 8-wave kernel: (halo row ty, col tx + dxi), k-chunk h
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int local = wm * 128 + mi * 32 + r;
    int ty, tx;
    if (g.pmode == 1) { const int win = local >> 2, wx = win % g.Wp; ty = 2 * (win / g.Wp) + ((local >> 1) & 1); tx = 2 * wx + (local & 1); }
    else if (g.pmode == 2) { const int win = local >> 1; tx = win % W; ty = 2 * (win / W) + (local & 1); }
    else { ty = local / W; tx = local - ty * W; }
#pragma unroll
    for (int dxi = 0; dxi < 3; ++dxi) {
      const int col = tx + dxi, sw = ((col >> 2) & 3) ^ ((ty & 1) << 1);
      abase[mi][dxi] = lbase + (unsigned)(ty * P + col) * 64u + (unsigned)((h ^ sw) << 4);
    }
  }
  const int swzb = (r >> 2) & 3;
  unsigned boff[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) boff[s2] = lbase + BRING + (wn * 128 + r) * 64 + (((2 * s2 + h) ^ swzb) << 4);

  // weight staging: rows (tid >> 2) + 64 j, position tid & 3 holds k-chunk (tid & 3) ^ ((tid >> 4) & 3).  N % 256 == 0: every row exists;
  // address = uniform base + 32-bit lane offset + 32-bit uniform k offset (one VALU add per piece, SGPR-base addressing); a step past the
  // end of K re-reads the last tile (never consumed) instead of selecting the zero page.
  const int srow = tid >> 2, bchunk = (tid & 3) ^ ((tid >> 4) & 3);
  const char* const wsrc = reinterpret_cast<const char*>(b.p);
  unsigned woff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) woff[j] = (unsigned)(((int64_t)(n_blk + 64 * j + srow) * b.ld + 8 * bchunk) * 2);
  unsigned char* const wbase = lds + wave * 1024;
  auto issue_b1 = [&](int step, unsigned kbytes, int j) {  // piece j (rows 64 j + 16 wave ..) of the weight tile of K step `step`
    dma16(wsrc + (woff[j] + kbytes), wbase + BRING + (step & 3) * BSLOT + j * 4096);
  };
  auto kof = [&](int step) -> unsigned { const int st = step < NT9 ? step : NT9 - 1, chunk = st / 9, tap = st - chunk * 9; return (unsigned)(tap * C + (chunk << 5)) * 2u; };
  // halo staging: this wave's ten pieces of a chunk (taps 0-4, two per step): group gq = 8 tap + 2 wave + e = 16 pixels of one halo row; lane ->
  // pixel (lane >> 2), position lane & 3.  Source pointer of chunk 0 per piece, 64 more bytes per chunk for the lanes inside the image; the
  // others (border, groups >= NG) stay on the zero page.  The last chunk re-reads its own halo into the other buffer (never consumed).
  const int hx = (lane >> 2) - 1, hchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const bf16_t* hsrc[10]; unsigned hinc = 0;              // bit i of hinc: lane is inside the image for piece i
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int gq = (i >> 1) * 8 + wave * 2 + (i & 1);
    const int row = gq / (P >> 4), col0 = (gq - row * (P >> 4)) << 4;
    const int y = y0 - 1 + row, x = col0 + hx;
    const bool ok = gq < NG && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    hsrc[i] = dma_select(ok, img + ((int64_t)(y * W + x) * C + ((hchunk ^ ((row & 1) << 1)) << 3)), zero);
    hinc |= ok ? (1u << i) : 0u;
  }
  auto issue_halo_i = [&](int i, int chunk, unsigned buf) {   // piece i of chunk `chunk`'s halo -> halo buffer at byte offset buf
    const int gq = (i >> 1) * 8 + wave * 2 + (i & 1);
    const unsigned inc = ((hinc >> i) & 1u) ? (unsigned)chunk << 6 : 0u;
    dma16(reinterpret_cast<const char*>(hsrc[i]) + inc, gq < NG ? lds + buf + gq * 1024 : lds + DUMP + wave * 1024);
  };
  auto issue_halo0 = [&](int gq) {                        // prologue: group gq of chunk 0
    const int row = gq / (P >> 4), col0 = (gq - row * (P >> 4)) << 4;
    const int y = y0 - 1 + row, x = col0 + hx;
    const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
    dma16(dma_select(ok, img + ((int64_t)(y * W + x) * C + ((hchunk ^ ((row & 1) << 1)) << 3)), zero), lds + gq * 1024);
  };

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // prologue: the whole halo of chunk 0 and three weight tiles
  for (int gq = wave; gq < NG; gq += 4) issue_halo0(gq);
#pragma unroll
  for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_b1(st, kof(st), j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  bf16x8 a0[4], b0[4], a1[4], b1[4];
#define AOCR_DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define AOCR_SB() __builtin_amdgcn_sched_barrier(0)
  // A fragment addresses: ab[mi][dxi] = abase + (halo buffer + dyi P 64) for the CURRENT (chunk, kernel row) -- twelve adds per three
  // steps -- and per read one XOR with the uniform (row-parity flip | k-half) bits, which sit below the 64-byte pixel pitch
  unsigned ab[4][3];
  auto set_ab = [&](unsigned hb, int kh) {
    const int dyi = SGN > 0 ? kh : 2 - kh;
    const unsigned U = hb + (unsigned)(dyi * P) * 64u;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) ab[mi][dxi] = abase[mi][dxi] + U;
  };
  auto a_addr = [&](int mi, int tap, int half) -> unsigned {   // tap, half: compile-time at every call site
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int dyi = SGN > 0 ? kh : 2 - kh, dxi = SGN > 0 ? kw : 2 - kw;
    const unsigned x = ((dyi & 1) << 5) ^ (half ? 32u : 0u);
    return x ? ab[mi][dxi] ^ x : ab[mi][dxi];
  };
  set_ab(0, 0);
  {                                                       // F0 of step 0
    AOCR_DSR(a0[0], a_addr(0, 0, 0), 0); AOCR_DSR(a0[1], a_addr(1, 0, 0), 0); AOCR_DSR(a0[2], a_addr(2, 0, 0), 0); AOCR_DSR(a0[3], a_addr(3, 0, 0), 0);
    AOCR_DSR(b0[0], boff[0], 0); AOCR_DSR(b0[1], boff[0], 2048); AOCR_DSR(b0[2], boff[0], 4096); AOCR_DSR(b0[3], boff[0], 6144);
  }
  unsigned long long pc0 = 0, pw0 = 0; if constexpr (TAG == 1) { pc0 = __builtin_readcyclecounter(); pw0 = wall_clock64(); }
  for (int chunk = 0; chunk < NC; ++chunk) {
    const unsigned hb = (chunk & 1) * HMAX, hbn = ((chunk + 1) & 1) * HMAX;
    const int chn = chunk + 1 < NC ? chunk + 1 : chunk;   // the chunk whose halo streams in (last chunk: its own again, into the other buffer)
    auto step_fn = [&](auto tapc) {
      constexpr int TAP = decltype(tapc)::value;
      constexpr int TN = (TAP + 1) % 9;                   // the next step's tap (chunk + 1 when TN == 0)
      constexpr int HT = TAP <= 4, HT1 = ((TAP + 8) % 9) <= 4, HT2 = ((TAP + 7) % 9) <= 4;
      constexpr int NVM = 6 + HT2 + 2 * HT1 + HT;
      const int step = chunk * 9 + TAP;
      constexpr int T3 = (TAP + 3) % 9, DC3 = (TAP + 3) / 9;      // (chunk, tap) of step s+3
      const unsigned k3 = chunk + DC3 < NC ? (unsigned)(T3 * C + ((chunk + DC3) << 5)) * 2u : (unsigned)(8 * C + ((NC - 1) << 5)) * 2u;
      const unsigned bs1 = (unsigned)((step & 3) * BSLOT), bs0n = (unsigned)(((step + 1) & 3) * BSLOT);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // F0(s), read under the MFMAs of the previous half
      AOCR_SB();
      // ---- half 1
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
          AOCR_SB();
          if (ni == 0) AOCR_DSR(a1[mi], a_addr(mi, TAP, 1), 0);
          if (ni == 2) { if (mi == 0) AOCR_DSR(b1[0], boff[1] + bs1, 0); else if (mi == 1) AOCR_DSR(b1[1], boff[1] + bs1, 2048); else if (mi == 2) AOCR_DSR(b1[2], boff[1] + bs1, 4096); else AOCR_DSR(b1[3], boff[1] + bs1, 6144); }
          if (ni == 3 && mi < 2) issue_b1(step + 3, k3, mi);
          if (ni == 3 && mi == 2 && HT) issue_halo_i(2 * TAP, chn, hbn);
          AOCR_SB();
        }
      }
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NVM) : "memory");   // own weight pieces of step s+1 landed; F1(s) back
      __builtin_amdgcn_s_barrier();                       // ... everyone's; everyone is done with weight slot s-1 and (last tap) with this chunk's first-half reads
      AOCR_SB();
      // ---- half 2 (its reads are the next step's first k-half: a new kernel row / chunk moves the A address table first)
      if (TN % 3 == 0) { set_ab(TN == 0 ? hbn : hb, TN / 3); AOCR_SB(); }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
          AOCR_SB();
          if (ni == 0) AOCR_DSR(a0[mi], a_addr(mi, TN, 0), 0);
          if (ni == 2) { if (mi == 0) AOCR_DSR(b0[0], boff[0] + bs0n, 0); else if (mi == 1) AOCR_DSR(b0[1], boff[0] + bs0n, 2048); else if (mi == 2) AOCR_DSR(b0[2], boff[0] + bs0n, 4096); else AOCR_DSR(b0[3], boff[0] + bs0n, 6144); }
          if (ni == 3 && mi < 2) issue_b1(step + 3, k3, 2 + mi);
          if (ni == 3 && mi == 2 && HT) issue_halo_i(2 * TAP + 1, chn, hbn);
          AOCR_SB();
        }
      }
    };
    step_fn(std::integral_constant<int, 0>{}); step_fn(std::integral_constant<int, 1>{}); step_fn(std::integral_constant<int, 2>{});
    step_fn(std::integral_constant<int, 3>{}); step_fn(std::integral_constant<int, 4>{}); step_fn(std::integral_constant<int, 5>{});
    step_fn(std::integral_constant<int, 6>{}); step_fn(std::integral_constant<int, 7>{}); step_fn(std::integral_constant<int, 8>{});
  }
#undef AOCR_DSR
#undef AOCR_SB
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the trailing (zero-page) pieces and reads must land before the LDS is released
  // the last half-step's reads of "the next step" are never consumed: keep their destination registers allocated up to here, or the compiler
  // hands them to the epilogue (seen: the destination pointer) while the LDS data is still in flight and lands on top of the new value
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a0[i]), "v"(b0[i]), "v"(a1[i]), "v"(b1[i]));
  if constexpr (TAG == 1) { if ((blockIdx.x == 17 || blockIdx.x == 300) && tid == 0) { const int o = blockIdx.x == 17 ? 4 : 0; g_kprobe[o] = __builtin_readcyclecounter() - pc0; g_kprobe[o + 1] = wall_clock64() - pw0; g_kprobe[o + 2] = pw0 - pe0; } }
  const int m0 = m_blk + wm * 128, n0 = n_blk + wn * 128;
  if (tile256_store_staged<4, 256>(ep, acc, lds, m_blk, n_blk, wm, wn, r, h, tid, opt)) {
    if constexpr (TAG == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if ((blockIdx.x == 17 || blockIdx.x == 300) && tid == 0) g_kprobe[blockIdx.x == 17 ? 7 : 3] = wall_clock64() - pe0; }
    return;
  }
  auto store_mi = [&](auto mic) {                         // compile-time mi: a rolled loop would index the accumulators dynamically (scratch)
    constexpr int mi = decltype(mic)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4][4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<4>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
  };
  store_mi(std::integral_constant<int, 0>{}); store_mi(std::integral_constant<int, 1>{});
  store_mi(std::integral_constant<int, 2>{}); store_mi(std::integral_constant<int, 3>{});
  if constexpr (TAG == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (blockIdx.x == 17 && tid == 0) g_kprobe[7] = wall_clock64() - pe0; }
}

// ---------------------------------------------------------------------------
// Narrow-N variant of the LDS-DMA kernel for the layers with 128 or 64 output columns (conv2 forward, the data
// gradients of conv2 / conv3): 256 x (64 NT) tiles, 8 waves as 4(M) x 2(N), 64 x (32 NT) outputs per wave.  The
// accumulators are small (32 NT VGPRs), so TWO workgroups share a CU -- one's epilogue (HBM-write bound) overlaps the
// other's K loop -- with a 3-slot ring per workgroup (tile t+2 issued under tile t).  Same source-side swizzle, zero
// page and barrier / vmcnt discipline as gemm_dma_bf16_kernel; for NT = 1 only waves 0-3 stage the 64-row B tile, so
// the per-tile piece count (and the vmcnt immediate) is wave-dependent.
// ---------------------------------------------------------------------------
template <class AL, class BL, class EP, int NT>
__global__ __launch_bounds__(512, 2)
void gemm_dma_narrow_kernel(AL a, BL b, EP ep, int K, int gx, int gy, const bf16_t* zero, int staged) {
  constexpr int BN = 64 * NT, SLOT = 16384 + BN * 64;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * SLOT];             // the ONLY LDS object
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * 256, n_blk = (bid % gx) * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;

  const int srow = tid >> 2, chunk = (tid & 3) ^ ((tid >> 4) & 3);
  const bool stage_b = NT == 2 || wave < 4;             // wave-uniform
  typename AL::DRow ra[2]; typename BL::DRow rb;
#pragma unroll
  for (int j = 0; j < 2; ++j) ra[j] = a.drow(m_blk + 128 * j + srow, chunk);
  rb = b.drow(n_blk + srow, chunk);                     // rows 0 .. BN-1 (threads past BN * 4 never issue it)
  typename AL::DCur ca = a.dseek(0); typename BL::DCur cb = b.dseek(0);
  const int nk = K >> 5;
  unsigned char* const wbase = lds + wave * 1024;
  int slot = 0;
  auto issue = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(a.dsrc(ra[j], ca, zero), wbase + slot * SLOT + j * 8192);
    a.dadvance(ca);
    if (stage_b) dma16(b.dsrc(rb, cb, zero), wbase + slot * SLOT + 16384);
    b.dadvance(cb);
    slot = slot == 2 ? 0 : slot + 1;
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int swz = (r >> 2) & 3;
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    aoff[s] = (wm * 64 + r) * 64 + (((2 * s + h) ^ swz) << 4);
    boff[s] = 16384 + (wn * 32 * NT + r) * 64 + (((2 * s + h) ^ swz) << 4);
  }

  issue(); issue();
  int rslot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (stage_b) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // this wave's pieces of tile kt have landed (tile kt+1 in flight)
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // ... everyone's have, and everyone is done reading tile kt-1
    const unsigned char* L = lds + rslot * SLOT;
    rslot = rslot == 2 ? 0 : rslot + 1;
    bf16x8 af[2][2], bf[2][NT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[s2][mi] = *reinterpret_cast<const bf16x8*>(L + aoff[s2] + mi * 2048);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) bf[s2][ni] = *reinterpret_cast<const bf16x8*>(L + boff[s2] + ni * 2048);
    }
    issue();                                            // tile kt+2 -> the slot of tile kt-1
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][mi], bf[s2][ni], acc[mi][ni], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (zero-page) tiles must land before the LDS is released
  if constexpr (std::is_same<EP, EpStore>::value) {
    if (staged && narrow_store_staged<NT, EP>(ep, acc, lds, m_blk, n_blk, wm, wn, r, h, tid)) return;
  }
  const int m0 = m_blk + wm * 64, n0 = n_blk + wn * 32 * NT;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[NT][4];
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<NT>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}

// ---------------------------------------------------------------------------
// 128 x 128 tiles on the LDS-DMA ring (round 4): the kernel for grids that do NOT fill the chip with 256 x 256 / 256 x 128 tiles -- the
// conv forward / data-gradient launches at 32-64 lines per GPU (BASELINE configs[3], the strong-scaling slice of C3) and ragged M.
// There gemm_lds_bf16_kernel -- global -> VGPR -> ds_write staging, ONE tile in flight, one workgroup per CU when the grid is <= 256 tiles --
// ran at 0.62 us per 128 x 128 x 32 step against 0.12 us of MFMA (conv6 at 32 lines: 90 us for 38.6 GFLOP): the step is the L2 round trip.
// Four waves of 64 x 64 (2 x 2 accumulator tiles), NS ring slots of 16 KB filled by LDS-DMA with NS - 1 tiles in flight (NS = 4: two
// workgroups per CU; NS = 8: one, for grids of <= one workgroup per CU), one barrier per tile, the same source-side swizzle, zero page and
// k order as the other LDS-DMA kernels (bit-identical to gemm_lds_bf16_kernel: the parity tests compare them under AOCR_NO_DMA128=1).
// ---------------------------------------------------------------------------
template <class AL, class BL, class EP, int NS, int NW>
__global__ __launch_bounds__(64 * NW, (NS == 4 && NW == 4) ? 2 : 1)
void gemm_dma128_kernel(AL a, BL b, EP ep, int K, int gx, int gy, const bf16_t* zero) {
  // NW = 4: waves 2 (M) x 2 (N), 64 x 64 each.  NW = 8: 2 x 4, 64 x 32 each -- for grids of <= one workgroup per CU: with ONE wave per SIMD the
  // wave's own LDS-DMA issue (4 pieces, ~80 cycles each), fragment reads and 8 MFMAs (256 cycles) serialise (measured 0.51 us per step, conv6 at
  // 32 lines 74 us); two waves per SIMD issue half the pieces each and multiply under each other's issue stalls.
  constexpr int SLOT = 16384, NI = NW == 4 ? 2 : 1, PPW = 16 / NW;      // N tiles per wave; DMA pieces per wave, tile and operand... (A and B: 8 pieces each)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * SLOT];            // the ONLY LDS object
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * 128, n_blk = (bid % gx) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = NW == 4 ? wave >> 1 : wave >> 2, wn = NW == 4 ? wave & 1 : wave & 3;

  // staging: rows (tid >> 2) + 16 NW j of both operands (j < PPW / 2 ... i.e. 128 rows over NW waves), position tid & 3 holds k-chunk (tid & 3) ^ ((tid >> 4) & 3)
  constexpr int NJ = 8 / NW;                            // pieces per wave and operand (a piece = 16 rows x 64 B)
  const int srow = tid >> 2, chunk = (tid & 3) ^ ((tid >> 4) & 3);
  typename AL::DRow ra[NJ]; typename BL::DRow rb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { ra[j] = a.drow(m_blk + 16 * NW * j + srow, chunk); rb[j] = b.drow(n_blk + 16 * NW * j + srow, chunk); }
  typename AL::DCur ca = a.dseek(0); typename BL::DCur cb = b.dseek(0);
  const int nk = K >> 5;
  unsigned char* const wbase = lds + wave * 1024;
  int slot = 0;
  auto issue = [&]() {                                  // (past the end of K the loaders select the zero page: never consumed)
#pragma unroll
    for (int j = 0; j < NJ; ++j) dma16(a.dsrc(ra[j], ca, zero), wbase + slot * SLOT + j * (1024 * NW));
    a.dadvance(ca);
#pragma unroll
    for (int j = 0; j < NJ; ++j) dma16(b.dsrc(rb[j], cb, zero), wbase + slot * SLOT + 8192 + j * (1024 * NW));
    b.dadvance(cb);
    slot = slot == NS - 1 ? 0 : slot + 1;
  };
  (void)PPW;

  f32x16 acc[2][NI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int swz = (r >> 2) & 3;
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    aoff[s] = (wm * 64 + r) * 64 + (((2 * s + h) ^ swz) << 4);
    boff[s] = 8192 + (wn * 32 * NI + r) * 64 + (((2 * s + h) ^ swz) << 4);
  }

#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue();
  int rslot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NJ * (NS - 2)) : "memory");   // this wave's pieces of tile kt have landed (NS - 2 tiles stay in flight)
    __builtin_amdgcn_s_barrier();                       // ... everyone's have, and everyone is done reading tile kt-1
    const unsigned char* L = lds + rslot * SLOT;
    rslot = rslot == NS - 1 ? 0 : rslot + 1;
    bf16x8 af[2][2], bf[2][NI];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) af[s2][mi] = *reinterpret_cast<const bf16x8*>(L + aoff[s2] + mi * 2048);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) bf[s2][ni] = *reinterpret_cast<const bf16x8*>(L + boff[s2] + ni * 2048);
    }
    issue();                                            // tile kt + NS - 1 -> the slot of tile kt-1
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][mi], bf[s2][ni], acc[mi][ni], 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (zero-page) tiles must land before the LDS is released
  const int m0 = m_blk + wm * 64, n0 = n_blk + wn * 32 * NI;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[NI][4];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * q + i];
      ep.template quad<NI>(m0 + 32 * mi + 8 * q + 4 * h, n0 + r, 32, v);
    }
}

// ---------------------------------------------------------------------------
// Filter-gradient kernel with hardware-transposed LDS reads.
// Both operands of dW = dY^T . Xcol are contiguous along M/N (channels) and strided along K (pixels).  Instead of
// transposing 8x4 micro-blocks in registers, the tiles are copied into LDS as they are -- [32 k][128 channels] bf16, one
// 16-byte piece per lane, 16 lanes = 256 contiguous bytes of one pixel row -- and the MFMA fragments (8 consecutive k
// of one channel) are gathered with ds_read_b64_tr_b16: per group of 16 lanes it reads a 4(k) x 16(channel) block and
// hands lane i the 4 k-values of channel i (probed on gfx950, tools/ubench/tr_probe.hip).  320-byte pitch: the 4 rows a
// 32-lane half touches start 16 dwords apart -> conflict-free.
// ---------------------------------------------------------------------------
constexpr int TR_PITCH = 320;
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}

template <class BL, class EP>
__device__ __forceinline__ void tr_tile(const LoadMNh& a, const BL& b, const EP& ep, int m_blk, int n_blk, int kbeg, int kend,
                                        unsigned char (&lds)[2][2][32 * TR_PITCH]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int nk = (kend - kbeg + 31) >> 5;

  // staging: thread -> k rows (tid>>4) and (tid>>4)+16, 16-byte piece (tid&15) = 8 channels
  const int krow = tid >> 4, piece = tid & 15;
  LoadMNh::Ctx8 ca = a.row8(m_blk + piece * 8);
  typename BL::Ctx8 cb = b.row8(n_blk + piece * 8);
  typename BL::Cur cur0 = b.seek(cb, kbeg + krow), cur1 = b.seek(cb, kbeg + krow + 16);
  int ka = kbeg + krow;
  const bf16_t* pa = ca.b + (int64_t)ka * a.ld;          // running pointers: no multiply per load
  const int64_t a16 = 16 * a.ld;
  uint4 ra[2], rb[2];
  auto gload = [&]() {
    ra[0] = (ca.ok && ka < a.K) ? *reinterpret_cast<const uint4*>(pa) : make_uint4(0, 0, 0, 0);
    ra[1] = (ca.ok && ka + 16 < a.K) ? *reinterpret_cast<const uint4*>(pa + a16) : make_uint4(0, 0, 0, 0);
    rb[0] = b.load8(cb, cur0); rb[1] = b.load8(cb, cur1);
    ka += 32; pa += 2 * a16; b.advance(cb, cur0); b.advance(cb, cur1);
  };
  auto lwrite = [&](int buf) {
    *reinterpret_cast<uint4*>(&lds[buf][0][krow * TR_PITCH + piece * 16]) = ra[0];
    *reinterpret_cast<uint4*>(&lds[buf][0][(krow + 16) * TR_PITCH + piece * 16]) = ra[1];
    *reinterpret_cast<uint4*>(&lds[buf][1][krow * TR_PITCH + piece * 16]) = rb[0];
    *reinterpret_cast<uint4*>(&lds[buf][1][(krow + 16) * TR_PITCH + piece * 16]) = rb[1];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read addressing: lane = 16g + 4q + p supplies row (.. + q), columns 16*(g&1) + 4p .. +3 of its 32-channel tile
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int tr_off = (8 * h + q) * TR_PITCH + (16 * (g & 1) + 4 * p) * 2;

  if (nk > 0) { gload(); lwrite(0); }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload();
    const unsigned char* la = &lds[buf][0][tr_off + (wm * 64) * 2];
    const unsigned char* lb = &lds[buf][1][tr_off + (wn * 64) * 2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        s16x4 a0 = lds_tr_read(la + (16 * s) * TR_PITCH + i * 64), a1 = lds_tr_read(la + (16 * s + 4) * TR_PITCH + i * 64);
        s16x4 b0 = lds_tr_read(lb + (16 * s) * TR_PITCH + i * 64), b1 = lds_tr_read(lb + (16 * s + 4) * TR_PITCH + i * 64);
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        af[i] = __builtin_bit_cast(bf16x8, av); bf[i] = __builtin_bit_cast(bf16x8, bv);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) lwrite(buf ^ 1);
    __syncthreads();
  }
  const int r = lane & 31;
  const int m0 = m_blk + wm * 64, n0 = n_blk + wn * 64;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * qq + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * qq + 4 * h, n0 + r, 32, v);
    }
}

template <class EP>
__global__ __launch_bounds__(256, 4) void conv_wgrad_tr_kernel(LoadMNh a, LoadConvXcolh b, EP ep, int K, int kper, int gx, int gy, int gz, float* part, long long pstride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][32 * TR_PITCH];
  const int nwg = gx * gy * gz, orig = blockIdx.x;                    // flat k-range-major order, see conv_wgrad_dma_kernel
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int zsp = lin / (gx * gy), bid = lin - zsp * (gx * gy);
  const int kbeg = zsp * kper;
  if (part) { ep.C = part + (size_t)zsp * pstride; ep.flags &= ~8 /* EP_ATOMIC (epilogues.h) */; }     // split-K slab of this k range: plain stores, summed by splitk_reduce
  tr_tile(a, b, ep, (bid / gx) * 128, (bid % gx) * 128, kbeg, min(K, kbeg + kper), lds);
}

// ---------------------------------------------------------------------------
// Filter gradient on the LDS-DMA skeleton of gemm_dma_bf16_kernel: 256 (Cout) x 256 (tap, ci) tiles, 8 waves, K = output
// pixels in 32-deep tiles through the 4-slot ring.  Both operands are contiguous along M/N, so a tile is [32 k][256
// channels] bf16 = 512 B per pixel row and a 1-KiB DMA piece is two pixel rows; fragments are gathered with
// ds_read_b64_tr_b16 as in conv_wgrad_tr_kernel.  No padding is possible in a DMA image, so the bank spread comes from
// an XOR on the source side: the 64-byte segment sg of pixel row k is stored at segment position sg ^ (k & 3); the four
// pixel rows one transposed read touches then sit in four different quarters of the 256-byte bank row.
// The im2col operand's piece addresses are per lane: (tap, ci) is a lane constant, the pixel cursor advances by 32.
// ---------------------------------------------------------------------------
template <class EP, int ABL = 0>                        // ABL: timing-only ablations (tools/ubench/wgrad_dma.hip): 1 no in-loop DMA, 2 no MFMA, 4 no fragment reads, 16 / 128 d y / x pieces from the zero page
__global__ __launch_bounds__(512, 1) void conv_wgrad_dma_kernel(LoadMNh a, LoadConvXcolh b, EP ep, int K, int kper, int gx, int gy, const bf16_t* zero, int gz, float* part, long long pstride) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 32768];          // the ONLY LDS object
  // (Round 1 gave every XCD its own K range with 36 tiles on its 32 CUs, i.e. two rounds: 1.5-2.5x slower.)
  // ONE flat grid over (k range, tile), renumbered so that the workgroups of an XCD are consecutive in k-range-major order: an
  // XCD's co-resident workgroups then share one (at most two) pixel ranges, so its L2 fetches d y and x of that range once for all
  // their tiles (before, with the k range on blockIdx.z, every XCD saw every range: 56 % L2 misses, 917 MB of fills per launch).
  // Worth 1 % on this kernel and 8 % on conv_wgrad_tr_kernel -- neither is bound by those misses.
  const int nwg = gx * gy * gz, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int zsp = lin / (gx * gy), bid = lin - zsp * (gx * gy);
  const int m_blk = (bid / gx) * 256, n_blk = (bid % gx) * 256;
  const int kbeg = zsp * kper, kend = min(K, kbeg + kper);
  // Split-K epilogue: float atomics execute at the memory side at ~1.3 TB/s chip-wide (66 MB of partial tiles = ~50 us of every
  // launch); with a slab per k range the partial tiles are plain stores (~6 TB/s) and one HBM-bound pass sums the slabs (splitk_reduce)
  if (part) { ep.C = part + (size_t)zsp * pstride; ep.flags &= ~8 /* EP_ATOMIC (epilogues.h) */; }
  const int nk = kend > kbeg ? (kend - kbeg + 31) >> 5 : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, wm = wave >> 2, wn = wave & 3;

  // staging role: pieces 2 wave, 2 wave + 1 of each operand; piece pi = pixel rows 2pi, 2pi+1; lane -> row 2pi + (lane>>5),
  // 16-byte position lane & 31 of the row, which holds logical chunk ((pos>>2) ^ (row & 3)) << 2 | (pos & 3)
  int krow[2], chunk[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    krow[j] = 2 * (2 * wave + j) + (lane >> 5);
    const int pos = lane & 31;
    chunk[j] = ((((pos >> 2) ^ (krow[j] & 3)) << 2) | (pos & 3));
  }
  LoadMNh::Ctx8 ca[2]; typename LoadConvXcolh::Ctx8 cb[2]; typename LoadConvXcolh::Cur cur[2]; int ka[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    ca[j] = a.row8(m_blk + 8 * chunk[j]); cb[j] = b.row8(n_blk + 8 * chunk[j]);
    cur[j] = b.seek(cb[j], kbeg + krow[j]); ka[j] = kbeg + krow[j];
  }
  const bf16_t* pa[2]; const int64_t astep = 32 * a.ld;   // running source pointers of the d y pieces (no multiply per piece and step)
#pragma unroll
  for (int j = 0; j < 2; ++j) pa[j] = ca[j].b + (int64_t)ka[j] * a.ld;
  unsigned char* const wbase = lds + (2 * wave) * 1024;
  int slot = 0;
  auto issue = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma16((ABL & 16) ? zero : dma_select(ca[j].ok && ka[j] < kend, pa[j], zero), wbase + slot * 32768 + j * 1024);
      ka[j] += 32; pa[j] += astep;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma16((ABL & 128) ? zero : b.dsrc8(cb[j], cur[j], kend, zero), wbase + slot * 32768 + 16384 + j * 1024);
      b.advance(cb[j], cur[j]);
    }
    slot = (slot + 1) & 3;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // transposed-read addressing (cf. tr_tile): lane = 16g + 4q + p supplies pixel row (.. + q), channels 16 (g&1) + 4p .. +3
  const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const unsigned rowoff = (8 * h + q) * 512 + (16 * (g & 1) + 4 * p4) * 2;
  unsigned aseg[4], bseg[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aseg[i] = rowoff + (((wm * 4 + i) ^ q) << 6);
#pragma unroll
  for (int i = 0; i < 2; ++i) bseg[i] = 16384 + rowoff + (((wn * 2 + i) ^ q) << 6);

  // The transposed reads are inline asm with a hand-placed lgkmcnt wait: behind the builtin (__builtin_amdgcn_ds_read_tr16_b64) hipcc
  // put `s_waitcnt vmcnt(0)` in front of the first read of every step -- it orders LDS reads behind ALL pending LDS-DMA -- which
  // drained the two tiles in flight, i.e. every step paid a full DMA round trip (found in the ISA; the kernel ran 1.37 us per
  // step against 1.04 for the forward kernel).  The wait asm takes the fragments as in/out operands so that no MFMA moves above it.
  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
#define AOCR_TRR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#pragma unroll
  for (int t = 0; t < 3; ++t) issue();
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // this wave's pieces of tile kt have landed (two later tiles in flight)
    __builtin_amdgcn_s_barrier();                       // ... everyone's have, and everyone is done reading tile kt-1
    const unsigned sl = lbase + (kt & 3) * 32768;
    u64 fa[2][4][2], fb[2][2][2];                       // [k half][tile][low / high four k]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ad = sl + aseg[i];
      if constexpr (ABL & 4) { fa[0][i][0] = fa[0][i][1] = fa[1][i][0] = fa[1][i][1] = ad; continue; }
      AOCR_TRR(fa[0][i][0], ad, 0); AOCR_TRR(fa[0][i][1], ad, 2048); AOCR_TRR(fa[1][i][0], ad, 8192); AOCR_TRR(fa[1][i][1], ad, 10240);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned ad = sl + bseg[i];
      if constexpr (ABL & 4) { fb[0][i][0] = fb[0][i][1] = fb[1][i][0] = fb[1][i][1] = ad; continue; }
      AOCR_TRR(fb[0][i][0], ad, 0); AOCR_TRR(fb[0][i][1], ad, 2048); AOCR_TRR(fb[1][i][0], ad, 8192); AOCR_TRR(fb[1][i][1], ad, 10240);
    }
    if constexpr (!(ABL & 1)) issue();                  // tile kt+3 -> the slot of tile kt-1
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[0][2][0]), "+v"(fa[0][2][1]),
                   "+v"(fa[0][3][0]), "+v"(fa[0][3][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1])
                 :: "memory");
    asm volatile(""
                 : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fa[1][2][0]), "+v"(fa[1][2][1]),
                   "+v"(fa[1][3][0]), "+v"(fa[1][3][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1])
                 :: "memory");
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const u64x2 av = {fa[s2][mi][0], fa[s2][mi][1]}, bv = {fb[s2][ni][0], fb[s2][ni][1]};
          if constexpr (ABL & 2) { acc[mi][ni][0] += (float)(unsigned)(av[0] ^ bv[1]); continue; }
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[mi][ni], 0, 0, 0);
        }
  }
#undef AOCR_TRR
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const int r = lane & 31;
  const int m0 = m_blk + wm * 128, n0 = n_blk + wn * 64;
  if constexpr (std::is_same<EP, EpStore>::value && ABL == 0) {
    // the slab tile of a k range is a plain fp32 store of a full 256 x 256 tile: through LDS as whole 1 KB rows (16-byte stores) instead of
    // 128 four-byte stores per lane
    if (part && !ep.flags && !ep.bias && !ep.bias2 && !ep.C1 && !ep.Cb && !ep.dg && m_blk + 256 <= ep.M && n_blk + 256 <= ep.N) {
      tile256_store_f32<2, 512>(ep.C, ep.ldc, nullptr, false, acc, lds, m_blk, n_blk, wm, wn, r, h, tid);
      return;
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      float v[2][4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ni][i] = acc[mi][ni][4 * qq + i];
      ep.template quad<2>(m0 + 32 * mi + 8 * qq + 4 * h, n0 + r, 32, v);
    }
}

// ---------------------------------------------------------------------------
// Round 4: the hoisted RECURRENT weight gradients (dW = dz^T x over all time steps: K = L B or T B rows, M, N = 256 .. 2048) on the same
// LDS-DMA skeleton, several problems per launch.  They ran on wgrad_tr_grouped_kernel (128 x 128 tiles, register staging, one tile in
// flight: 111 GFLOP in 377 us = 0.12 of the bf16 peak at C3) on the side stream beside the encoder BPTT -- and that stream had become the
// critical path: the main stream idled 121 us at the join in front of the CNN backward pass (tools/ktrace_step.sh c3).
// Both operands are plain [K][M] / [K][N] bf16 matrices (row strides lda / ldb), so the B side loses conv_wgrad_dma_kernel's tap / pixel
// cursor; a work item is (problem, 256 x 256 tile, k range) and writes its fp32 tile into the problem's slab of that k range (whole 1 KB
// rows through LDS); wgrad_slab_reduce_kernel adds the slabs into the (strided) gradient matrices.  M, N multiples of 256, K of 32.
// ---------------------------------------------------------------------------
struct WgDmaProblem { const bf16_t* A; const bf16_t* B; long long lda, ldb; int M, N, K, gx, tiles, ks, kper, first; float* part; float* C; long long ldc; long long f4first; };
struct WgDmaArgs { int n, total; long long f4total; WgDmaProblem p[8]; };
template <int UNUSED = 0>      // (a template only so that the header may be included by several translation units)
__global__ __launch_bounds__(512, 1) void wgrad_dma_grouped_kernel(WgDmaArgs g, const bf16_t* zero) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 32768];          // the ONLY LDS object
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.p[i].first) pi = i;
  const WgDmaProblem& P = g.p[pi];
  const int local = blockIdx.x - P.first;                     // k-range-major inside a problem: co-resident workgroups share a row range of both operands
  const int zsp = local / P.tiles, bid = local - zsp * P.tiles;
  const int m_blk = (bid / P.gx) * 256, n_blk = (bid % P.gx) * 256;
  const int kbeg = zsp * P.kper, kend = min(P.K, kbeg + P.kper);
  const int nk = kend > kbeg ? (kend - kbeg + 31) >> 5 : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, wm = wave >> 2, wn = wave & 3;
  // staging role (as conv_wgrad_dma_kernel): pieces 2 wave, 2 wave + 1 of each operand; piece pi = k rows 2pi, 2pi+1; lane -> row 2pi + (lane>>5),
  // 16-byte position lane & 31 of the row, which holds logical chunk ((pos>>2) ^ (row & 3)) << 2 | (pos & 3)
  const bf16_t* pa[2]; const bf16_t* pb[2]; int ka[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int krow = 2 * (2 * wave + j) + (lane >> 5), pos = lane & 31;
    const int chunk = ((((pos >> 2) ^ (krow & 3)) << 2) | (pos & 3));
    ka[j] = kbeg + krow;
    pa[j] = P.A + (long long)ka[j] * P.lda + m_blk + 8 * chunk; pb[j] = P.B + (long long)ka[j] * P.ldb + n_blk + 8 * chunk;
  }
  const long long astep = 32 * P.lda, bstep = 32 * P.ldb;
  unsigned char* const wbase = lds + (2 * wave) * 1024;
  int slot = 0;
  auto issue = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) { dma16(dma_select(ka[j] < kend, pa[j], zero), wbase + slot * 32768 + j * 1024); pa[j] += astep; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { dma16(dma_select(ka[j] < kend, pb[j], zero), wbase + slot * 32768 + 16384 + j * 1024); pb[j] += bstep; ka[j] += 32; }
    slot = (slot + 1) & 3;
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int gq = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const unsigned rowoff = (8 * h + q) * 512 + (16 * (gq & 1) + 4 * p4) * 2;
  unsigned aseg[4], bseg[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aseg[i] = rowoff + (((wm * 4 + i) ^ q) << 6);
#pragma unroll
  for (int i = 0; i < 2; ++i) bseg[i] = 16384 + rowoff + (((wn * 2 + i) ^ q) << 6);
  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
#define AOCR_TRR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
#pragma unroll
  for (int t = 0; t < 3; ++t) issue();
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // this wave's pieces of tile kt have landed (two later tiles in flight)
    __builtin_amdgcn_s_barrier();
    const unsigned sl = lbase + (kt & 3) * 32768;
    u64 fa[2][4][2], fb[2][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ad = sl + aseg[i];
      AOCR_TRR(fa[0][i][0], ad, 0); AOCR_TRR(fa[0][i][1], ad, 2048); AOCR_TRR(fa[1][i][0], ad, 8192); AOCR_TRR(fa[1][i][1], ad, 10240);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned ad = sl + bseg[i];
      AOCR_TRR(fb[0][i][0], ad, 0); AOCR_TRR(fb[0][i][1], ad, 2048); AOCR_TRR(fb[1][i][0], ad, 8192); AOCR_TRR(fb[1][i][1], ad, 10240);
    }
    issue();                                            // tile kt+3 -> the slot of tile kt-1
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[0][2][0]), "+v"(fa[0][2][1]),
                   "+v"(fa[0][3][0]), "+v"(fa[0][3][1]), "+v"(fb[0][0][0]), "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1])
                 :: "memory");
    asm volatile(""
                 : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fa[1][2][0]), "+v"(fa[1][2][1]),
                   "+v"(fa[1][3][0]), "+v"(fa[1][3][1]), "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]), "+v"(fb[1][1][1])
                 :: "memory");
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const u64x2 av = {fa[s2][mi][0], fa[s2][mi][1]}, bv = {fb[s2][ni][0], fb[s2][ni][1]};
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[mi][ni], 0, 0, 0);
        }
  }
#undef AOCR_TRR
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  tile256_store_f32<2, 512>(P.part + (size_t)zsp * P.M * P.N, P.N, nullptr, false, acc, lds, m_blk, n_blk, wm, wn, lane & 31, h, tid);
}
// C[m][n] (row stride ldc) += sum over the k ranges of the problem's slabs; one launch for all problems of a group (float4 items)
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(WgDmaArgs g) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.f4total) return;
  int pi = 0;
#pragma unroll
  for (int k = 1; k < 8; ++k) if (k < g.n && i >= g.p[k].f4first) pi = k;
  const WgDmaProblem& P = g.p[pi];
  const long long li = i - P.f4first; const int n4 = P.N >> 2; const long long m = li / n4; const int c = (int)(li - m * n4) * 4;
  const size_t mn = (size_t)P.M * P.N;
  const float* src = P.part + (size_t)m * P.N + c;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < P.ks; ++z) acc += __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + (size_t)z * mn));
  float* o = P.C + m * P.ldc + c;                               // (ldc need not be a multiple of 4: dW_i2h[:, E:] starts at column E)
  o[0] += acc[0]; o[1] += acc[1]; o[2] += acc[2]; o[3] += acc[3];
}

// ---------------------------------------------------------------------------
// Filter gradient of a 3 x 3 / pad 1 layer with the INPUT MAP HALO-RESIDENT (round 4; priced in round 3).
//   dW[co][tap][ci] = sum over pixels p of  dy[p][co] . x[p + off(tap)][ci]
// conv_wgrad_dma_kernel tiles N as ONE tap x 256 input channels, so every K step (32 pixels) moves 16 KB of d y AND 16 KB of tap-shifted x through
// L2 -> LDS, and nine workgroups re-fetch the same x pixels for the nine taps.  Here an N tile is ALL NINE TAPS x 32 input channels (N = 288):
// a K step = one row segment of 32 pixels stages d y as before (16 KB) plus the 3 x 34-pixel halo of the segment for the 32 channels (6.4 KB), and
// a tap is an OFFSET of the transposed fragment read inside that halo -- 22.4 KB instead of 32 KB per step, for 12 % more MFMA work per step.
// Four waves, each 64 output channels x 288 (2 x 9 accumulator tiles of 32 x 32 = 288 AGPRs), one wave per SIMD.
//   LDS slot (24 KB): [32 pixel rows][256 co] d y image as in conv_wgrad_dma_kernel (512 B per pixel row, 64-byte segment index ^ (row & 3)), then the
//   halo [3 rows][34 pixels][32 ci] = 64 B per pixel, LINEAR: the transposed read of a 16-lane group takes 4 consecutive pixels x 32 channels = 256
//   contiguous bytes -- conflict-free at every tap shift without a swizzle.  The halo is filled by 7 LDS-DMA pieces whose lanes address
//   (halo row, pixel, 16-byte chunk) slots in image order (outside the map / past slot 407: zero page); the source offset is LINEAR in the segment
//   index (raster pixels), so a piece's pointer advances by a constant and only the two bounds compares change per step.
// Grid: (Cout / 256) x (Cin / 32) tiles x split-K over whole segments, k-range-major XCD order (the workgroups of an XCD share pixel ranges in its L2);
// every k range writes its 256 x 288 partial tile with plain stores into its slab (dW layout [Cout][9 Cin]); splitk_reduce sums the slabs.
// A K step never straddles an image row: rows of W % 32 != 0 pixels end with a ragged segment whose pixel slots past W are zero in both operands (round 6: the
// reference-default shape's 25- / 50-wide maps, the width buckets of C4).  Cout % 256 == 0 (128: MG = 2), Cin % 32 == 0.
// ---------------------------------------------------------------------------
template <int... I, class F> __device__ __forceinline__ void aocr_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void aocr_static_for(F&& f) { aocr_static_for_impl(std::make_integer_sequence<int, N>{}, f); }
template <int TAG = 0, int MG = 4, bool RG = false, bool IM = false>       // RG: ragged rows (W % 32 != 0); the whole-segment form keeps its constant pointer steps and has no validity compare on d y.  IM: the DMA issue between the two k-halves' MFMAs, which start as soon as their own fragments are back (AOCR_NO_WGRAD_ISSUE_MID=1: the one-wait form)
__global__ __launch_bounds__(128 * MG, 1)
void conv_wgrad_halo_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ part, long long pstride,
                            int nimg, int H, int W, int Cin, int Cout, int gx, int gy, int gz, int segs_per, const bf16_t* zero) {
  // EIGHT waves as 4 (output-channel groups of 64) x 2 (tap groups: taps 0-4 / taps 5-8): a wave holds 2 x 5 or 2 x 4 accumulator tiles (160 / 128
  // registers), and since waves w and w + 4 share a SIMD every SIMD carries 18 MFMAs per k-half.  History of the shape (tools/ubench/wgrad_halo.hip,
  // conv6, random operands, conv_wgrad_dma_kernel 293-327 us in the same harness):
  //   four waves of 64 x 288: 288 accumulator registers exceed the 256 AGPRs hipcc gives a kernel, and it moves the two tiles it keeps in VGPRs through
  //     AGPRs around every MFMA (200 v_accvgpr_mov per half step): 380 us compiler-scheduled, 491 us with hand-placed reads / DMA pieces;
  //   eight waves of 32 x 288 (every wave re-reads all nine taps' fragments: 2.2 transposed reads per MFMA): 276 us isolated -- but only 5 % faster than
  //     conv_wgrad_dma_kernel back to back on the real gradients and no faster inside the step: the LDS -> register bytes it adds cost what the L2 -> LDS
  //     bytes it saves;
  //   this form: 1.4 transposed reads per MFMA (conv_wgrad_dma_kernel: 1.5) AND 22.4 instead of 32 KB per step through L2 -> LDS.
  // MG = 4: tiles of 256 output channels, eight waves.  MG = 2 (layers with 128 output channels: conv2): tiles of 128, four waves as 2 x 2, one per SIMD.
  constexpr int MT = 64 * MG, NWV = 2 * MG, AROW = MT * 2, RPP = 1024 / AROW, LPR = 64 / RPP;   // tile rows (output channels), waves, bytes per pixel row of the d y image, pixel rows / lanes per row of a 1 KB piece
  constexpr int BOFF = 32 * AROW, SLOT = BOFF + 8192, NS = 6, PW = 34, HPW = 8 / NWV;          // halo pieces per wave
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NWV * 20480 > NS * SLOT ? NWV * 20480 : NS * SLOT];   // the ONLY LDS object: ring of NS slots; epilogue: NWV waves x 32 rows x 160 fp32
  const int nwg = gx * gy * gz, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int zsp = lin / (gx * gy), bid = lin - zsp * (gx * gy);
  const int m_blk = (bid / gx) * MT, cchunk = bid % gx;                            // MT output channels x input channels [32 cchunk, +32)
  const int spr = (W + 31) >> 5, S = nimg * H * spr;                               // segments per row (round 6: the last one of a row may be ragged -- pixel slots past W are zero), in all
  const int s_beg = zsp * segs_per, s_end = min(S, s_beg + segs_per);
  const int nk = s_end > s_beg ? s_end - s_beg : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, mg = wave & (MG - 1), tg = wave / MG;

  // ---- d y staging: pieces 2 wave, 2 wave + 1; piece pi = pixel rows 2 pi, 2 pi + 1; lane -> row 2 pi + (lane >> 5), 16-byte position lane & 31 of the
  // row, which holds logical chunk ((pos >> 2) ^ (row & 3)) << 2 | (pos & 3)   (as conv_wgrad_dma_kernel)
  // (segment s = (image row s / spr, position s % spr) starts at raster pixel (s / spr) W + 32 (s % spr): the pointers advance by 32 pixels inside a row and by
  //  what is left of the row, W - 32 (spr - 1), at its end -- both 32 where W % 32 == 0)
  const int wrap_px = W - 32 * (spr - 1);
  const int64_t seg0_px = (int64_t)(s_beg / spr) * W + 32 * (s_beg % spr);
  const bf16_t* pa[2]; int akrow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int krow = RPP * (2 * wave + j) + lane / LPR, pos = lane & (LPR - 1);
    const int chunk = ((((pos >> 2) ^ (krow & 3)) << 2) | (pos & 3));
    akrow[j] = krow;
    pa[j] = dy + (seg0_px + krow) * Cout + m_blk + 8 * chunk;
  }
  const int64_t astep = (int64_t)32 * Cout, awrap = (int64_t)wrap_px * Cout;
  // ---- halo staging: piece `wave` (piece 7 is all padding); slot sl = 64 piece + lane -> (halo row ry, pixel cx, chunk c) in image order
  bool bin[HPW]; int bry[HPW], bcx[HPW]; const bf16_t* pb[HPW];
#pragma unroll
  for (int j = 0; j < HPW; ++j) {
    const int sl = 64 * (HPW * wave + j) + lane;
    bin[j] = sl < 3 * PW * 4;
    const int hry = sl / (PW * 4), hrem = sl - hry * (PW * 4), hcx = hrem >> 2, hc = hrem & 3;
    bry[j] = hry - 1; bcx[j] = hcx - 1;
    pb[j] = x + (seg0_px + (int64_t)(hry - 1) * W + (hcx - 1)) * Cin + cchunk * 32 + hc * 8;       // never dereferenced while outside the map
  }
  const int64_t bstep = (int64_t)32 * Cin, bwrap = (int64_t)wrap_px * Cin;
  int is = s_beg, ixs = s_beg % spr, iy = (s_beg / spr) % H;                       // the issue stream's segment: index, position in its row, image row
  unsigned char* const wA = lds + (2 * wave) * 1024;
  unsigned char* const wB = lds + BOFF + (HPW * wave) * 1024;
  int islot = 0;
  auto issue = [&]() {
    const bool live = is < s_end;
    const int x0 = ixs << 5;
    if constexpr (RG) {
      const bool last = ixs + 1 == spr;                 // (uniform) the row ends with this segment
#pragma unroll
      for (int j = 0; j < 2; ++j) { dma16(dma_select(live && x0 + akrow[j] < W, pa[j], zero), wA + islot * SLOT + j * 1024); pa[j] += last ? awrap : astep; }
#pragma unroll
      for (int j = 0; j < HPW; ++j) {
        const bool ok = live && bin[j] && (unsigned)(iy + bry[j]) < (unsigned)H && (unsigned)(x0 + bcx[j]) < (unsigned)W;
        dma16(dma_select(ok, pb[j], zero), wB + islot * SLOT + j * 1024); pb[j] += last ? bwrap : bstep;
      }
      ++is; if (last) { ixs = 0; if (++iy == H) iy = 0; } else ++ixs;
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) { dma16(dma_select(live, pa[j], zero), wA + islot * SLOT + j * 1024); pa[j] += astep; }
#pragma unroll
      for (int j = 0; j < HPW; ++j) {
        const bool ok = live && bin[j] && (unsigned)(iy + bry[j]) < (unsigned)H && (unsigned)(x0 + bcx[j]) < (unsigned)W;
        dma16(dma_select(ok, pb[j], zero), wB + islot * SLOT + j * 1024); pb[j] += bstep;
      }
      ++is; if (++ixs == spr) { ixs = 0; if (++iy == H) iy = 0; }
    }
    islot = islot == NS - 1 ? 0 : islot + 1;
  };

  // transposed-read addressing: lane = 16 g + 4 q + p supplies pixel row 8 h + q (+ 4), channels 16 (g & 1) + 4 p .. + 3
  const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
  const unsigned rowoff = lbase + (8 * h + q) * AROW + (16 * (g & 1) + 4 * p4) * 2;
  const unsigned aseg0 = rowoff + (((2 * mg) ^ q) << 6), aseg1 = rowoff + (((2 * mg + 1) ^ q) << 6);
  const unsigned brd = lbase + BOFF + (8 * h + q) * 64 + (16 * (g & 1) + 4 * p4) * 2;
  typedef unsigned long long u64;
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const int r = lane & 31;
  float* const slab = part + (size_t)zsp * pstride;
  const int64_t ldw = (int64_t)9 * Cin;
#define AOCR_TRH(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))

  issue(); issue(); issue();
  auto body = [&](auto tgc) {                           // TG = 0: taps 0 .. 4, TG = 1: taps 5 .. 8
    constexpr int TG = decltype(tgc)::value, T0 = TG ? 5 : 0, NTP = TG ? 4 : 5;
    f32x16 acc[2][NTP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    int rslot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (2 + HPW)) : "memory");  // this wave's pieces of segment kt have landed (two later segments in flight)
      __builtin_amdgcn_s_barrier();                     // ... everyone's have, and everyone is done reading segment kt-1
      const unsigned so = rslot * SLOT;
      rslot = rslot == NS - 1 ? 0 : rslot + 1;
      u64 fa[2][2][2], fb[2][NTP][2];                   // [k half][tile / tap][low / high four k]
      const unsigned a0 = aseg0 + so, a1 = aseg1 + so, bd = brd + so;
      AOCR_TRH(fa[0][0][0], a0, 0); AOCR_TRH(fa[0][0][1], a0, 4 * AROW); AOCR_TRH(fa[0][1][0], a1, 0); AOCR_TRH(fa[0][1][1], a1, 4 * AROW);
#define AOCR_TAPH(S2, T) do { AOCR_TRH(fb[S2][T][0], bd, (((T0 + T) / 3) * PW + (T0 + T) % 3) * 64 + S2 * 1024); AOCR_TRH(fb[S2][T][1], bd, (((T0 + T) / 3) * PW + (T0 + T) % 3) * 64 + S2 * 1024 + 256); } while (0)
      AOCR_TAPH(0, 0); AOCR_TAPH(0, 1); AOCR_TAPH(0, 2); AOCR_TAPH(0, 3); if constexpr (NTP == 5) AOCR_TAPH(0, NTP - 1);
      AOCR_TRH(fa[1][0][0], a0, 16 * AROW); AOCR_TRH(fa[1][0][1], a0, 20 * AROW); AOCR_TRH(fa[1][1][0], a1, 16 * AROW); AOCR_TRH(fa[1][1][1], a1, 20 * AROW);
      AOCR_TAPH(1, 0); AOCR_TAPH(1, 1); AOCR_TAPH(1, 2); AOCR_TAPH(1, 3); if constexpr (NTP == 5) AOCR_TAPH(1, NTP - 1);
#undef AOCR_TAPH
      auto mma_half = [&](auto s2c) {
        constexpr int s2 = decltype(s2c)::value;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int t = 0; t < NTP; ++t) {
            const u64x2 av = {fa[s2][mi][0], fa[s2][mi][1]}, bv = {fb[s2][t][0], fb[s2][t][1]};
            acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[mi][t], 0, 0, 0);
          }
      };
      if constexpr (!IM) {
        issue();                                        // segment kt + 3 -> a slot last read three steps ago
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]) :: "memory");
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int t = 0; t < NTP; ++t) asm volatile("" : "+v"(fb[s2][t][0]), "+v"(fb[s2][t][1]));
        mma_half(std::integral_constant<int, 0>{}); mma_half(std::integral_constant<int, 1>{});
      } else {
        // Round 6: the wave issues in order, and with the two waves of a SIMD behind one barrier per step everything in front of a wave's first MFMA is time the
        // matrix pipe stands still.  The first k-half's MFMAs now start as soon as ITS 4 + 2 NTP fragment reads are back (LDS returns in order: the second half's
        // reads, issued behind them, land underneath), and the address work + DMA issue of segment kt + 3 sits between the halves.  Same products, same order.
        // (Measured beside it: the loop rotated so that the next segment's first-half reads also land under MFMAs -- no better: the 28 transposed reads of a
        //  wave and step are an LDS-bandwidth cost, 115 KB per CU and step, not a latency.)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]) : "n"(4 + 2 * NTP) : "memory");
#pragma unroll
        for (int t = 0; t < NTP; ++t) asm volatile("" : "+v"(fb[0][t][0]), "+v"(fb[0][t][1]));
        __builtin_amdgcn_sched_barrier(0);
        mma_half(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        issue();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]) :: "memory");
#pragma unroll
        for (int t = 0; t < NTP; ++t) asm volatile("" : "+v"(fb[1][t][0]), "+v"(fb[1][t][1]));
        mma_half(std::integral_constant<int, 1>{});
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                                    // every wave is out of the K loop: the ring is free
    // ---- epilogue: the wave's 64 x (32 NTP) partial tile into its slab, 32 rows at a time through the wave's own 20 KB of LDS, as 16-byte stores of
    // 128-byte runs (the 32 input channels of one tap of one output channel)
    unsigned char* const wl = lds + wave * 20480;
    constexpr int RP = NTP * 128;                       // bytes per staged row
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int t = 0; t < NTP; ++t)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float*>(wl + (8 * qq + 4 * h + i) * RP + (t * 32 + r) * 4) = acc[mi][t][4 * qq + i];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the wave reads back only what it wrote itself)
      const int co0 = m_blk + mg * 64 + mi * 32;
#pragma unroll 4
      for (int it = 0; it < 4 * NTP; ++it) {            // 32 rows x 8 NTP float4
        const int idx = it * 64 + lane, row = idx / (8 * NTP), c4 = idx - row * (8 * NTP), t = c4 >> 3, j = c4 & 7;
        const float4 v = *reinterpret_cast<const float4*>(wl + row * RP + c4 * 16);
        *reinterpret_cast<float4*>(slab + (int64_t)(co0 + row) * ldw + (int64_t)(T0 + t) * Cin + cchunk * 32 + j * 4) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  };
  if (tg == 0) body(std::integral_constant<int, 0>{}); else body(std::integral_constant<int, 1>{});
#undef AOCR_TRH
}

// Grouped form: up to 8 independent contractions of the same operand kinds in ONE launch (the hoisted weight gradients
// of a recurrent stack are each too small to fill the chip; together their tiles do, without split-K atomics).
template <class AL, class BL, class EP> struct GroupProblem { AL a; BL b; EP ep; int K, kper, gx, ksplit, first; };
template <class AL, class BL, class EP> struct GroupArgs { int n, total; GroupProblem<AL, BL, EP> p[8]; };

template <class AL, class BL, class EP>
__global__ __launch_bounds__(256) void gemm_lds_grouped_kernel(GroupArgs<AL, BL, EP> g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][128 * LDS_PITCH];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.p[i].first) pi = i;
  const GroupProblem<AL, BL, EP>& P = g.p[pi];
  const int local = blockIdx.x - P.first;                   // tile-major, then k slice
  const int tile = local / P.ksplit, z = local - tile * P.ksplit;
  const int kbeg = z * P.kper;
  lds_tile<32>(P.a, P.b, P.ep, (tile / P.gx) * 128, (tile % P.gx) * 128, kbeg, min(P.K, kbeg + P.kper), lds);
}
// the same grouping over bf16 shadows with transposed LDS reads (A_i [K][M], B_i [K][N] bf16)
template <class EP>
__global__ __launch_bounds__(256, 4) void wgrad_tr_grouped_kernel(GroupArgs<LoadMNh, LoadMNh, EP> g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][32 * TR_PITCH];
  int pi = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.p[i].first) pi = i;
  const GroupProblem<LoadMNh, LoadMNh, EP>& P = g.p[pi];
  const int local = blockIdx.x - P.first;
  const int tile = local / P.ksplit, z = local - tile * P.ksplit;
  const int kbeg = z * P.kper;
  tr_tile(P.a, P.b, P.ep, (tile / P.gx) * 128, (tile % P.gx) * 128, kbeg, min(P.K, kbeg + P.kper), lds);
}

// ---------------------------------------------------------------------------
// "small" kernel for the recurrent steps (M = batch): one 32 x (32*NT) output tile per
// block, K split over the block's 4 waves and reduced through LDS.  With GATES the NT=4
// tiles are the four gate blocks of the same 32 hidden units: B row = g*gate_stride + j.
// blockIdx.z selects one of two argument sets (the two encoder directions).
// ---------------------------------------------------------------------------
template <class AL, class BL, class EP> struct SmallArgs { AL a; BL b; EP ep; int K; };

// QG (quarter gate tiles, NT = 1): the ONE 32-column tile of a workgroup carries all four gates of 8 hidden units (column j = gate
// j >> 3 of unit n0 + (j & 7)), so a small batch still spreads over H / 8 workgroups per row block instead of H / 32 (C2, batch 64:
// 128 instead of 32 workgroups per decoder layer and step); the epilogue gathers a unit's four gate values from lanes j, j+8, j+16, j+24.
template <bool BF16, int NT, bool GATES, int NW, bool QG, class AL, class BL, class EP>
__device__ __forceinline__ void gemm_small_body(const SmallArgs<AL, BL, EP>& g, int gate_stride, float* red) {
  constexpr int NV = Mode<BF16>::NV, CH = Mode<BF16>::CHUNK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32;
  const int n0 = QG ? blockIdx.x * 8 : (GATES ? blockIdx.x * 32 : blockIdx.x * 32 * NT);
  const int K = g.K;

  typename AL::Ctx ca[1]; typename BL::Ctx cb[NT];
  ca[0] = g.a.row(m0 + r);
#pragma unroll
  for (int i = 0; i < NT; ++i) cb[i] = g.b.row(QG ? (r >> 3) * gate_stride + n0 + (r & 7) : (GATES ? i * gate_stride + n0 + r : n0 + 32 * i + r));

  f32x16 acc[1][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;

  typedef typename FragOf<AL, NV>::type FA; typedef typename FragOf<BL, NV>::type FB;
  // The step kernels are latency-bound (one workgroup per CU, a serial chain of L2 round trips), so each wave keeps
  // D chunks in flight: a ring of D fragment sets, refilled right after the MFMAs that consumed the slot.
  // Each wave owns a CONTIGUOUS quarter of K and its D in-flight chunks are consecutive, so the 16-32 bytes a lane
  // takes from its row per chunk add up to whole 128-byte lines inside the in-flight window (row-strided fragment
  // loads with chunks interleaved across waves re-fetched every line several times through the 32 KB L1).
  constexpr int D = 4;
  const int kw = ((K + NW * CH - 1) / (NW * CH)) * CH;      // K range per wave, multiple of CH
  const int kend = min(K, (wave + 1) * kw);
  FA fa[D][1]; FB fb[D][NT];
  int kb = wave * kw;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int kc = kb + d * CH;
    if (kc < kend) {
      g.a.template load_fast<NV>(fa[d][0], ca[0], kc + NV * h);
#pragma unroll
      for (int i = 0; i < NT; ++i) g.b.template load_fast<NV>(fb[d][i], cb[i], kc + NV * h);
    }
  }
  for (; kb < kend; kb += D * CH) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int kc = kb + d * CH;
      if (kc < kend) {
        mma_chunk<BF16, 1, NT, FA, FB>(acc, fa[d], fb[d]);
        const int kn = kc + D * CH;
        if (kn < kend) {
          g.a.template load_fast<NV>(fa[d][0], ca[0], kn + NV * h);
#pragma unroll
          for (int i = 0; i < NT; ++i) g.b.template load_fast<NV>(fb[d][i], cb[i], kn + NV * h);
        }
      }
    }
  }
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[((wave * NT + ni) * 16 + e) * 64 + lane] = acc[0][ni][e];
  __syncthreads();
  if (NW > 4 && wave >= 4) return;          // waves 4.. only contribute partial sums
  const int q = wave;                       // this wave finalises rows 8q+4h .. +3
  float v[NT][4];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[((w * NT + ni) * 16 + 4 * q + i) * 64 + lane];
      v[ni][i] = s;
    }
  if constexpr (QG) {
    float v4[4][4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
      for (int i = 0; i < 4; ++i) v4[gt][i] = __shfl(v[0][i], (lane & 32) + 8 * gt + (r & 7), 64);
    if (r < 8) g.ep.template quad<4>(m0 + 8 * q + 4 * h, n0 + r, 32, v4);
  } else {
    g.ep.template quad<NT>(m0 + 8 * q + 4 * h, n0 + r, 32, v);
  }
}

template <class AL, class BL, class EP> struct SmallArgs2 { SmallArgs<AL, BL, EP> z[3]; };     // up to three problems per launch (blockIdx.z)

template <bool BF16, int NT, bool GATES, class AL, class BL, class EP, int NW = 4, bool QG = false>
__global__ __launch_bounds__(64 * NW) void gemm_small_kernel(SmallArgs2<AL, BL, EP> zz, int gate_stride) {
  __shared__ float red[NW * NT * 16 * 64];
  gemm_small_body<BF16, NT, GATES, NW, QG>(zz.z[blockIdx.z], gate_stride, red);  // kernarg array: one body, scalar-indexed
}

// ---------------------------------------------------------------------------
// bf16 recurrent-step kernel with WAVE-PRIVATE LDS staging.
// gemm_small_kernel loads MFMA fragments straight from global memory: every wave-instruction then touches 32-64
// different 128-byte lines and uses 16-32 bytes of each, and the CU's vector memory path (about one line per clock)
// -- not bytes, not MFMA -- sets the time (measured ~37 GB/s per CU).  Here each wave streams its K quarter in
// 64-deep chunks with full-line loads (A fp32: 4 rows x 256 B per instruction, B bf16 shadows: 8 rows x 128 B),
// converts/copies them into its own LDS image (144-byte pitch: conflict-free ds_read_b128 fragments) and feeds the
// MFMAs from there.  No workgroup barrier inside the K loop (LDS serves a wave's requests in order); the next
// chunk's global loads are issued before the current chunk's MFMAs.  Requires K, K0 % 64 == 0, N % 32 == 0.
// ---------------------------------------------------------------------------
constexpr int STEP_PITCH = 144;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NT, int NW = 4, int MT = 1> constexpr int step_lds_bytes() { return NW * (MT + NT) * 32 * STEP_PITCH; }

// global -> registers for one 64-deep chunk (kept as free functions with flat, statically indexed arrays of native
// vector types: arrays captured by reference in lambdas / HIP's uint4 struct were demoted to LDS or scratch)
template <int NT, int GATES>
__device__ __forceinline__ void step_gload_b(const LoadKh2& b, int kc, int K0, int brow0, int gate_stride, int bp, u32x4 (&rb)[NT * 4]) {
  const bool s1 = kc >= K0;
  const bf16_t* pb = s1 ? b.p1 : b.p0; const int64_t ldb = s1 ? b.ld1 : b.ld0;
  const int kk = s1 ? kc - K0 : kc;
#pragma unroll
  for (int t = 0; t < NT * 4; ++t) {
    const int ni = t >> 2, i = t & 3;
    // GATES == 2 (half tiles): tile ni, tile row q = br + 8i -> gate 2ni + (q >> 4) = 2ni + (i >> 1), hidden unit n0 + (q & 15)
    const int row = GATES == 2 ? (2 * ni + (i >> 1)) * gate_stride + brow0 + 8 * (i & 1)
                               : (GATES ? ni * gate_stride : 32 * ni) + brow0 + 8 * i;
    rb[t] = *reinterpret_cast<const u32x4*>(pb + (int64_t)row * ldb + kk + 8 * bp);
  }
}
// A from fp32 (8 x dwordx4: 4 rows x 256 B per instruction) ...
__device__ __forceinline__ void step_gload_a(const LoadK& a, int kc, const int (&arow)[8], int ap, float4 (&ra)[8]) {
  const bool s1 = kc >= a.K0;
  const float* pa = s1 ? a.p1 : a.p0; const int64_t lda = s1 ? a.ld1 : a.ld0;
  const int kk = s1 ? kc - a.K0 : kc;
#pragma unroll
  for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const float4*>(pa + (int64_t)arow[i] * lda + kk + 4 * ap);
}
// ... or from its bf16 shadow (4 x dwordx4: 8 rows x 128 B per instruction)
__device__ __forceinline__ void step_gload_a(const LoadKh2& a, int kc, const int (&arow)[4], int bp, u32x4 (&ra)[4]) {
  const bool s1 = kc >= a.K0;
  const bf16_t* pa = s1 ? a.p1 : a.p0; const int64_t lda = s1 ? a.ld1 : a.ld0;
  const int kk = s1 ? kc - a.K0 : kc;
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const u32x4*>(pa + (int64_t)arow[i] * lda + kk + 8 * bp);
}
__device__ __forceinline__ void step_lwrite_a(unsigned char* la, int ar, int ap, const float4 (&ra)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bf16x4 v; v[0] = (bf16_t)ra[i].x; v[1] = (bf16_t)ra[i].y; v[2] = (bf16_t)ra[i].z; v[3] = (bf16_t)ra[i].w;
    *reinterpret_cast<bf16x4*>(la + (ar + 4 * i) * STEP_PITCH + ap * 8) = v;
  }
}
__device__ __forceinline__ void step_lwrite_a(unsigned char* la, int br, int bp, const u32x4 (&ra)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(la + (br + 8 * i) * STEP_PITCH + bp * 16) = ra[i];
}
template <int NT>
__device__ __forceinline__ void step_lwrite_b(unsigned char* lb, int br, int bp, const u32x4 (&rb)[NT * 4]) {
#pragma unroll
  for (int t = 0; t < NT * 4; ++t)
    *reinterpret_cast<u32x4*>(lb + ((t >> 2) * 32 + br + 8 * (t & 3)) * STEP_PITCH + bp * 16) = rb[t];
}

// GATES: 0 plain (NT column tiles of 32); 1 gate tiles (tile = gate, 32 hidden units per workgroup, NT = 4); 2 HALF gate tiles
// (NT = 2: tile t carries gates 2t and 2t+1 of 16 hidden units in its column halves, so a workgroup loads and multiplies half
// the weights and twice as many workgroups share the step: 128 -> 256 at Hd = 512, B = 256; the epilogue pairs lanes l and l^16).
// NW waves share K (4, or 8 where the LDS allows): one workgroup per CU means the waves of ONE workgroup are all the memory-level
// parallelism a CU has, and these launches are a load -> MFMA -> reduce latency chain.
// MT (round 5): row tiles of 32 per workgroup.  MT = 2 at large M (the reference-default decoder: M = 400 rows, N = 4 x 1024 gate columns, K = 2048): every B (weight)
// fragment read from LDS feeds two MFMAs and the weight tile is streamed by 7 row blocks instead of 13 -- these launches run at the chip's L2 -> CU rate
// (266 MB per launch at ~7 TB/s), so bytes per launch is what counts.
template <int NT, int GATES, class AL, class EP, int NW = 4, int MT = 1>
__global__ __launch_bounds__(64 * NW) void gemm_step_kernel(SmallArgs2<AL, LoadKh2, EP> zz, int gate_stride) {
  constexpr bool AH = SrcBf16<AL>::v;                           // A operand read from its bf16 shadow
  constexpr int NA = AH ? 4 : 8;
  constexpr int E = 16 / NW;                                    // accumulator rows finished per thread
  __shared__ __attribute__((aligned(16))) unsigned char lds[step_lds_bytes<NT, NW, MT>()];
  const SmallArgs<AL, LoadKh2, EP>& g = zz.z[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32 * MT;
  const int n0 = GATES == 2 ? blockIdx.x * 16 : (GATES ? blockIdx.x * 32 : blockIdx.x * 32 * NT);
  const int K = g.K;
  unsigned char* la = lds + wave * ((MT + NT) * 32 * STEP_PITCH);
  unsigned char* lb = la + MT * 32 * STEP_PITCH;
  const int kw = ((K / 64 + NW - 1) / NW) * 64;             // this wave's K range (multiple of 64)
  const int kbeg = wave * kw, kend = min(K, kbeg + kw);

  // staging roles.  fp32 rows: lane -> row (lane>>4) + 4i, 16-byte piece (lane&15) = 4 k;  bf16 rows: row (lane>>3) + 8i, piece (lane&7) = 8 k
  const int ar = AH ? (lane >> 3) : (lane >> 4), ap = AH ? (lane & 7) : (lane & 15), br = lane >> 3, bp = lane & 7;
  int arow[MT][NA];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < NA; ++i) arow[mt][i] = min(m0 + 32 * mt + ar + (AH ? 8 : 4) * i, g.a.rows - 1);   // rows past the end: any valid row, result dropped
  const int brow0 = n0 + br;                                  // N % 32 == 0 so always valid

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][j][e] = 0.f;

  // the epilogue's own operands (zx, c_prev, gates, ...) are requested now so that they arrive during the K loop
  // accumulator index i = E*wave + e of a 32x32 tile sits in row 8*(i/4) + 4h + i%4
  const int orow = m0 + 8 * ((E * wave) >> 2) + 4 * h + ((E * wave) & 3);
  typename EP::Pre pre[MT][E];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < E; ++e) pre[mt][e] = g.ep.prefetch(orow + 32 * mt + e, GATES == 2 ? n0 + (r & 15) : n0 + r);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int e = 0; e < E; ++e) g.ep.prefetch_zx(pre[mt][e], orow + 32 * mt + e, GATES == 2 ? n0 + (r & 15) : n0 + r);       // (second half: rows looked up through the first half's tokens)

  if (kbeg < kend) {
    typename std::conditional<AH, u32x4, float4>::type ra[MT][NA]; u32x4 rb[NT * 4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) step_gload_a(g.a, kbeg, arow[mt], ap, ra[mt]);
    step_gload_b<NT, GATES>(g.b, kbeg, g.a.K0, brow0, gate_stride, bp, rb);
    for (int kc = kbeg; kc < kend; kc += 64) {
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) step_lwrite_a(la + mt * 32 * STEP_PITCH, ar, ap, ra[mt]);
      step_lwrite_b<NT>(lb, br, bp, rb);
      __builtin_amdgcn_wave_barrier();
      if (kc + 64 < kend) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) step_gload_a(g.a, kc + 64, arow[mt], ap, ra[mt]);
        step_gload_b<NT, GATES>(g.b, kc + 64, g.a.K0, brow0, gate_stride, bp, rb);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = *reinterpret_cast<const bf16x8*>(la + (mt * 32 + r) * STEP_PITCH + 32 * s + 16 * h);
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(lb + (ni * 32 + r) * STEP_PITCH + 32 * s + 16 * h);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr, acc[mt][ni], 0, 0, 0);
        }
      }
    }
  }
  // cross-wave reduction: every wave parks its accumulators in its own (now idle) LDS region -- one row tile at a time (the region holds NT tiles of 4 KB)
  constexpr int WSTRIDE = (MT + NT) * 32 * STEP_PITCH / 4;    // floats between two waves' regions
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
  if (mt > 0) __syncthreads();                                // every thread is done reading the tile before
  __builtin_amdgcn_wave_barrier();
  float* red = reinterpret_cast<float*>(la);
#pragma unroll
  for (int ni = 0; ni < NT; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) red[(ni * 16 + e) * 64 + lane] = acc[mt][ni][e];
  __syncthreads();
  // each thread finishes E elements (rows orow + e of column r) straight from LDS
  const float* r0 = reinterpret_cast<const float*>(lds);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    float v[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const int idx = (ni * 16 + E * wave + e) * 64 + lane;
      v[ni] = (r0[idx] + r0[idx + WSTRIDE]) + (r0[idx + 2 * WSTRIDE] + r0[idx + 3 * WSTRIDE]);
#pragma unroll
      for (int q = 4; q < NW; q += 4)
        v[ni] += (r0[idx + q * WSTRIDE] + r0[idx + (q + 1) * WSTRIDE]) + (r0[idx + (q + 2) * WSTRIDE] + r0[idx + (q + 3) * WSTRIDE]);
    }
    if constexpr (GATES == 2) {                                 // lane l < 16 of each 32-lane half: [i, o] here, [f, g] in lane l + 16
      float v4[4] = {v[0], __shfl_xor(v[0], 16, 64), v[1], __shfl_xor(v[1], 16, 64)};
      if (r < 16) g.ep.template elem<4>(orow + 32 * mt + e, n0 + r, 32, v4, pre[mt][e]);
    } else {
      g.ep.template elem<NT>(orow + 32 * mt + e, n0 + r, 32, v, pre[mt][e]);
    }
  }
  }
}

// ---------------------------------------------------------------------------
// Round 4: the EXACT-fp32 recurrent-step kernel with wave-private LDS staging (the fp32 sibling of gemm_step_kernel).  In fp32 mode
// (BASELINE configs[1]) the recurrent steps ran on gemm_small_kernel<false>, whose lanes load their MFMA fragments straight from global
// memory -- 16 bytes of 32 different rows per wave instruction, ~37 GB/s per CU: 13-30 us per launch at batch 64, 4.2 of the step's
// 9.5 ms.  Here each of the NW waves streams its contiguous share of K in 32-deep chunks with whole 128-byte lines (8 rows x 128 B per
// load instruction, two chunks in flight), parks them in its own fp32 LDS image (144-byte pitch: conflict-free ds_read_b128) and feeds
// v_mfma_f32_32x32x2_f32 from there; no workgroup barrier inside the K loop.  One 32 x 32 output tile per workgroup (NT = 1):
//   QG = true : gate tiles -- tile column j = gate j >> 3 of hidden unit n0 + (j & 7) (8 units per workgroup, as gemm_small_kernel's QG)
//   QG = false: 32 plain columns.
// Requires K % 32 == 0, K0 % 32 == 0 for a two-segment operand (a chunk never straddles the segments), 16-byte aligned rows, N % 32 == 0 (plain) / H % 8 == 0 (QG).
// ---------------------------------------------------------------------------
struct StepfSrc { const float* p0; const float* p1; int64_t ld0, ld1; int K0; };      // one operand: a buffer or two K segments ([x0 ; x1]); scalars (SGPRs)
__device__ __forceinline__ void stepf_gload(const StepfSrc& a, const StepfSrc& b, int kc, const int (&arow)[4], const int (&brow)[4], int sp, f32x4 (&xa)[4], f32x4 (&xb)[4]) {
  const bool sa1 = kc >= a.K0, sb1 = kc >= b.K0;                 // kc is wave-uniform: scalar selects
  const float* pa = sa1 ? a.p1 : a.p0; const int64_t lda = sa1 ? a.ld1 : a.ld0;
  const float* pb = sb1 ? b.p1 : b.p0; const int64_t ldb = sb1 ? b.ld1 : b.ld0;
  const int ka = (sa1 ? kc - a.K0 : kc) + 4 * sp, kb = (sb1 ? kc - b.K0 : kc) + 4 * sp;
#pragma unroll
  for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const f32x4*>(pa + (int64_t)arow[i] * lda + ka);
#pragma unroll
  for (int i = 0; i < 4; ++i) xb[i] = *reinterpret_cast<const f32x4*>(pb + (int64_t)brow[i] * ldb + kb);
}
__device__ __forceinline__ void stepf_lwrite(unsigned char* la, unsigned char* lb, int sr, int sp, const f32x4 (&xa)[4], const f32x4 (&xb)[4]) {
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(la + (sr + 8 * i) * STEP_PITCH + sp * 16) = xa[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(lb + (sr + 8 * i) * STEP_PITCH + sp * 16) = xb[i];
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void stepf_mma(const unsigned char* la, const unsigned char* lb, int r, int h, f32x16& acc) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 af = *reinterpret_cast<const f32x4*>(la + r * STEP_PITCH + 32 * c + 16 * h);
    const f32x4 bf = *reinterpret_cast<const f32x4*>(lb + r * STEP_PITCH + 32 * c + 16 * h);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], acc, 0, 0, 0);
  }
}
template <bool QG, class EP, int NW = 8>
__global__ __launch_bounds__(64 * NW) void gemm_step_f32_kernel(SmallArgs2<LoadK, LoadK, EP> zz, int gate_stride) {
  constexpr int E = 16 / NW, PITCH = STEP_PITCH, WBYTES = 2 * 32 * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NW * WBYTES];      // 8 waves: 73.7 KB
  const SmallArgs<LoadK, LoadK, EP>& g = zz.z[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (scalar: the K range and the segment selects stay in SGPRs)
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32;
  const int n0 = QG ? blockIdx.x * 8 : blockIdx.x * 32;
  const int K = g.K;
  unsigned char* const la = lds + wave * WBYTES;
  unsigned char* const lb = la + 32 * PITCH;
  const int kw = ((K / 32 + NW - 1) / NW) * 32;                  // this wave's K range (multiple of 32)
  const int kbeg = wave * kw, kend = min(K, kbeg + kw);
  const int sr = lane >> 3, sp = lane & 7;                       // staging: tile rows sr + 8 i, 16-byte piece sp of the 128-byte chunk row
  int arow[4], brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    arow[i] = min(m0 + sr + 8 * i, g.a.rows - 1);                // rows past the end: any valid row, result dropped by the epilogue
    brow[i] = min(QG ? i * gate_stride + n0 + sr : n0 + sr + 8 * i, g.b.rows - 1);
  }
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int orow = m0 + 8 * ((E * wave) >> 2) + 4 * h + ((E * wave) & 3);
  typename EP::Pre pre[E];
#pragma unroll
  for (int e = 0; e < E; ++e) pre[e] = g.ep.prefetch(orow + e, QG ? n0 + (r & 7) : n0 + r);
#pragma unroll
  for (int e = 0; e < E; ++e) g.ep.prefetch_zx(pre[e], orow + e, QG ? n0 + (r & 7) : n0 + r);

  // (free functions over flat arrays of native vector types: float4 arrays handed to lambdas were demoted to scratch.)  Every chunk's loads are
  // issued UNCONDITIONALLY -- past the wave's range they re-read its last chunk -- so that the number of loads in flight is the same on every
  // path into the loop: with conditional prefetches the compiler's vmcnt at the loop head assumed the shortest path and waited for everything.
  if (kbeg < kend) {
    const StepfSrc A = {g.a.p0, g.a.p1 ? g.a.p1 : g.a.p0, g.a.ld0, g.a.ld1, g.a.K0}, B = {g.b.p0, g.b.p1 ? g.b.p1 : g.b.p0, g.b.ld0, g.b.ld1, g.b.K0};
    const int klast = kend - 32;
    f32x4 a0[4], b0[4], a1[4], b1[4];
    stepf_gload(A, B, kbeg, arow, brow, sp, a0, b0);
    stepf_gload(A, B, min(kbeg + 32, klast), arow, brow, sp, a1, b1);
    int kc = kbeg;
    for (; kc + 64 <= kend; kc += 64) {                                       // pairs of chunks: no condition inside, so the counts at the loop head are exact
      stepf_lwrite(la, lb, sr, sp, a0, b0);
      stepf_gload(A, B, min(kc + 64, klast), arow, brow, sp, a0, b0);        // two chunks ahead: lands during this chunk's and the next chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      stepf_mma(la, lb, r, h, acc);
      stepf_lwrite(la, lb, sr, sp, a1, b1);
      stepf_gload(A, B, min(kc + 96, klast), arow, brow, sp, a1, b1);
      __builtin_amdgcn_sched_barrier(0);
      stepf_mma(la, lb, r, h, acc);
    }
    if (kc < kend) {                                                          // odd chunk count: the last chunk is in a0 / b0
      stepf_lwrite(la, lb, sr, sp, a0, b0);
      stepf_mma(la, lb, r, h, acc);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                       // vmcnt(0): the re-read tail chunks are not left in flight across the reduction
  }
  // cross-wave reduction: every wave parks its accumulators in its own (now idle) LDS region
  __builtin_amdgcn_wave_barrier();
  float* red = reinterpret_cast<float*>(la);
#pragma unroll
  for (int e = 0; e < 16; ++e) red[e * 64 + lane] = acc[e];
  __syncthreads();
  const float* r0 = reinterpret_cast<const float*>(lds);
  constexpr int WSTRIDE = WBYTES / 4;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int idx = (E * wave + e) * 64 + lane;
    float v = (r0[idx] + r0[idx + WSTRIDE]) + (r0[idx + 2 * WSTRIDE] + r0[idx + 3 * WSTRIDE]);
#pragma unroll
    for (int q = 4; q < NW; q += 4)
      v += (r0[idx + q * WSTRIDE] + r0[idx + (q + 1) * WSTRIDE]) + (r0[idx + (q + 2) * WSTRIDE] + r0[idx + (q + 3) * WSTRIDE]);
    if constexpr (QG) {                                           // lane r < 8 gathers the four gates of unit n0 + r from lanes r, r + 8, r + 16, r + 24
      float v4[4];
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) v4[gt] = __shfl(v, (lane & 32) + 8 * gt + (r & 7), 64);
      if (r < 8) g.ep.template elem<4>(orow + e, n0 + r, 32, v4, pre[e]);
    } else {
      const float v1[1] = {v};
      g.ep.template elem<1>(orow + e, n0 + r, 32, v1, pre[e]);
    }
  }
}

}  // namespace aocr
