// dec_cluster.hip -- the teacher-forced decoder loop (model.lua:553-568 train, :604-627 gold pass; cell LSTM.lua:18-122, attention
// LSTM.lua:124-162) as ONE launch for all L steps: a GROUP of 32 compute units (one XCD under round-robin dispatch) owns 32 batch
// rows, and every recurrent weight the loop needs -- W1 = [W1_i2h(feed part) | W1_h2h], W2 = [W2_i2h | W2_h2h], W_a, W_c: 10 MB in bf16
// at Hd = 512 -- is RESIDENT IN THE REGISTERS of the group (288 VGPRs per lane: member m owns hidden units 16m .. 16m+15 of both
// layers and output columns 16m .. 16m+15 of W_a / W_c).  The launch chain this replaces runs 5 dependent kernels per step, each of
// which re-streams its weights through L2 and pays a launch + drain (~8 us each at C3: 0.94 ms for 24 steps).
//
// Per step, four all-gathers of a 32 x 512 bf16 operand inside the group (8-byte {2 x bf16, tag} granules: rnn_cluster.hip):
//   out(t-1) -> [feed ; h1(t-1)] W1^T + zx1(t) -> gates -> c1, h1          (zx1 = embedding part + biases, hoisted over all L steps)
//   h1(t)    -> [h1(t) ; h2(t-1)] W2^T + b -> gates -> c2, h2
//   h2(t)    -> attention of row r on member r (q = W_a h2, which only the backward pass reads, is one GEMM over all L steps after the loop):
//               s = ctxA[r] . h2  (ctxA = ctx . W_a precomputed once: ctx . (W_a h) = (ctx W_a) . h), a = softmax(s), c = a . ctx[r]
//   c(t)     -> out = tanh(W_c [c ; h2])
// Operands live in LDS ([32 rows][512] bf16 x 3 buffers); the weights of a wave are MFMA A fragments (transposed products, as in
// rnn_cluster.hip), tile rows ordered [unit][gate] so that a lane holds the four gates of one (unit, batch row) cell.
// Everything the backward pass reads (gates, states, q, a, [c ; h2], out; fp32 + bf16 shadows) is written in the layouts of the
// launch chain, which stays the fallback (fp32 mode, Hd != 512, other layer counts, no input feed).
#include "ops.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace aocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

namespace {
constexpr int DC_SPIN_LIMIT = 1 << 18;
constexpr int HD = 512, NM = 32, R = 32, PA = HD * 2 + 16;      // hidden size, members per group, rows per group, LDS operand pitch

__device__ __forceinline__ void dpoll16(u32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void dwait0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dpin(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void dst_granules(void* p, u32x4 v, bool local) {
  if (local) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned bfbits(float x) { bf16_t h = (bf16_t)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
__device__ __forceinline__ u64 ldg64(const u64* p) { return __hip_atomic_load(const_cast<u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ddma4(const void* g, unsigned char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}

// all-gather of one 32 x 512 bf16 operand into an LDS buffer: element (row, 4-unit chunk c) is the pair of granules at
// src + row * RS + (c >> 2) * MS + 2 * (c & 3); 16 polls of 16 bytes per thread.  Returns false after a timeout.
template <int RS, int MS>
__device__ __forceinline__ bool gather(const u64* src, unsigned tag, unsigned char* dst, int tid, int* err, int code) {
  bool dead = false;
  __syncthreads();                              // every wave of this workgroup is past its reads of the previous contents of dst
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    u32x4 gr[8];
    auto issue = [&]() {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const int idx = tid + 256 * (8 * half + j), row = idx >> 7, ch = idx & 127; dpoll16(gr[j], src + (size_t)row * RS + (ch >> 2) * MS + 2 * (ch & 3)); }
    };
    issue(); dwait0();
#pragma unroll
    for (int j = 0; j < 8; ++j) dpin(gr[j]);
    int spins = 0;
    while (!dead) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 8; ++j) ok = ok && gr[j][1] == tag && gr[j][3] == tag;
      if (__all(ok)) break;
      if (++spins > DC_SPIN_LIMIT) { dead = true; if ((tid & 63) == 0) atomicExch(err, code); break; }
      __builtin_amdgcn_s_sleep(1);
      issue(); dwait0();
#pragma unroll
      for (int j = 0; j < 8; ++j) dpin(gr[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int idx = tid + 256 * (8 * half + j), row = idx >> 7, ch = idx & 127; *reinterpret_cast<u32x2*>(dst + (size_t)row * PA + ch * 8) = u32x2{gr[j][0], gr[j][2]}; }
  }
  return !dead;
}
// the same operand from a plain bf16 array [B][512] (step 0: the initial states)
__device__ __forceinline__ void load_rows(const bf16_t* src, int row0, int B, unsigned char* dst, int tid) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = tid + 256 * j, row = idx >> 6, ch = idx & 63;                          // 64 chunks of 8 bf16 per row
    const int gr = min(row0 + row, B - 1);
    *reinterpret_cast<u32x4*>(dst + (size_t)row * PA + ch * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)gr * HD + ch * 8);
  }
}
}  // namespace

__global__ __launch_bounds__(256, 1) void dec_cl_fwd_kernel(DecClFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const F = lds;                        // feed (out(t-1)), later c(t)
  unsigned char* const H1 = lds + R * PA;              // h1(t-1), later h1(t)
  unsigned char* const H2 = lds + 2 * R * PA;          // h2(t-1), later h2(t)
  unsigned char* const zxs = lds + 3 * R * PA;         // [4 waves][2 rt][4 gates][256 B]: zx1 staging (LDS-DMA)
  float* const red = reinterpret_cast<float*>(lds + 3 * R * PA + 8192);     // [4 waves][2][64][4] partial tiles / attention scratch (8 KB)
  float* const sc = red + 2048;                        // [<= 256] attention scores / probabilities, then [4][512] partial context (8 KB + 1 KB)
  __shared__ int s_local, s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % NM, gl = (i8 / NM) * 8 + xcd;
  if (gl >= p.ngroups) return;
  const int group = p.group0 + gl;
  const int B = p.B, T = p.T, L = p.L, row0 = group * R;
  const int unit = 16 * member + 4 * wave + q;                   // gate epilogues: this lane's hidden unit (lane = (batch row c16, unit))
  const int arow_u = 16 * member + 4 * wave + (c16 >> 2), arow_g = c16 & 3;      // A fragment row: tile row c16 = 4 * unit + gate

  // ---- co-location check (rnn_cluster.hip): plain granule stores are only visible to the group's polls inside one XCD
  u64* const xt = p.xtab + (size_t)group * NM;
  if (tid == 0) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u;
    stg64(xt + member, ((u64)p.epoch << 32) | (u64)(xcc + 1u));
    int same = 1;
    for (int m = 0; m < NM; ++m) {
      u64 v; int spins = 0;
      while ((unsigned)((v = ldg64(xt + m)) >> 32) != p.epoch) { if (++spins > DC_SPIN_LIMIT) { atomicExch(p.err, 13); same = 0; break; } __builtin_amdgcn_s_sleep(2); }
      if ((unsigned)v != xcc + 1u) same = 0;
    }
    s_local = same && !p.force_remote; s_dead = 0;
  }
  __syncthreads();
  const bool local = __builtin_amdgcn_readfirstlane(s_local) != 0;

  // ---- resident weights (A fragments)
  bf16x8 w1[32], w2[32], wcr[8];
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    const size_t ro = (size_t)(arow_g * HD + arow_u) * HD + 32 * (s & 15) + 8 * q;
    w1[s] = *reinterpret_cast<const bf16x8*>((s < 16 ? p.w1i : p.w1h) + ro);
    w2[s] = *reinterpret_cast<const bf16x8*>((s < 16 ? p.w2i : p.w2h) + ro);
  }
#pragma unroll
  for (int s = 0; s < 8; ++s) wcr[s] = *reinterpret_cast<const bf16x8*>(p.wc + (size_t)(16 * member + c16) * 2 * HD + 256 * wave + 32 * s + 8 * q);
  float b2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b2[i] = p.b2i[i * HD + unit] + p.b2h[i * HD + unit];
  float c1[2], c2[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) { const int row = min(row0 + 16 * rt + c16, B - 1); c1[rt] = p.cs[0][(size_t)row * HD + unit]; c2[rt] = p.cs[1][(size_t)row * HD + unit]; }

  u64* const xg = p.xbuf + (size_t)group * 4 * 2 * 8192;          // [kind: out, h1, h2, c][parity][8192 granules]
  auto dma_zx = [&](int t) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = min(row0 + 16 * rt + c16, B - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) ddma4(p.zx1 + ((size_t)t * B + row) * 4 * HD + i * HD + unit, zxs + wave * 2048 + (rt * 4 + i) * 256);
    }
  };
  dma_zx(0);
  load_rows(p.out_b, row0, B, F, tid); load_rows(p.hsb[0], row0, B, H1, tid); load_rows(p.hsb[1], row0, B, H2, tid);
  dwait0();
  __syncthreads();
  bool dead = false;

  for (int t = 0; t < L && !dead; ++t) {
    const unsigned tagc = p.epoch * 4096u + (unsigned)(t + 1);    // tag of everything produced in step t
    const int par = t & 1;
    const size_t slot = (size_t)B * HD;
    // =================== layer 1: z1 = [feed ; h1(t-1)] W1^T + zx1(t)
    if (t > 0) {
      const bool okg = gather<8, 256>(xg + (size_t)(0 * 2 + ((t - 1) & 1)) * 8192, p.epoch * 4096u + (unsigned)t, F, tid, p.err, 11);
      if (!okg) s_dead = 1;
      __syncthreads();
      if (s_dead) { dead = true; break; }
    }
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const unsigned char* src = s < 16 ? F : H1;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt + c16) * PA + ((s & 15) * 32 + 8 * q) * 2);
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[s], bv, acc[rt], 0, 0, 0);
        }
      }
      unsigned hb[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = row0 + 16 * rt + c16;
        float z[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = acc[rt][i] + *reinterpret_cast<const float*>(zxs + wave * 2048 + (rt * 4 + i) * 256 + lane * 4);
        const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), og = sigmoidf_(z[2]), gg = tanhf_(z[3]);
        const float cn = fg * c1[rt] + ig * gg, hn = og * tanhf_(cn);
        c1[rt] = cn; hb[rt] = bfbits(hn);
        if (row < B) {
          if (p.gates[0]) { float* gp = p.gates[0] + ((size_t)t * B + row) * 4 * HD + unit; gp[0] = ig; gp[HD] = fg; gp[2 * HD] = og; gp[3 * HD] = gg; }
          const size_t o = (size_t)(t + 1) * slot + (size_t)row * HD + unit;
          p.cs[0][o] = cn; p.hs[0][o] = hn; p.hsb[0][o] = (bf16_t)hn;
        }
      }
      // publish h1(t): the four units of (row, wave) are in lanes q = 0..3 of column c16 -> lane q = 0 packs them
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const unsigned h1v = __shfl(hb[rt], lane + 16, 64), h2v = __shfl(hb[rt], lane + 32, 64), h3v = __shfl(hb[rt], lane + 48, 64);
        if (q == 0) dst_granules(xg + (size_t)(1 * 2 + par) * 8192 + ((size_t)(member * 32 + 16 * rt + c16) * 8 + 2 * wave), u32x4{hb[rt] | (h1v << 16), tagc, h2v | (h3v << 16), tagc}, local);
      }
    }
    // =================== layer 2: z2 = [h1(t) ; h2(t-1)] W2^T + b
    {
      const bool okg = gather<8, 256>(xg + (size_t)(1 * 2 + par) * 8192, tagc, H1, tid, p.err, 12);
      if (!okg) s_dead = 1;
      __syncthreads();
      if (s_dead) { dead = true; break; }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int s = 0; s < 32; ++s) {
        const unsigned char* src = s < 16 ? H1 : H2;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt + c16) * PA + ((s & 15) * 32 + 8 * q) * 2);
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[s], bv, acc[rt], 0, 0, 0);
        }
      }
      unsigned hb[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = row0 + 16 * rt + c16;
        const float ig = sigmoidf_(acc[rt][0] + b2[0]), fg = sigmoidf_(acc[rt][1] + b2[1]), og = sigmoidf_(acc[rt][2] + b2[2]), gg = tanhf_(acc[rt][3] + b2[3]);
        const float cn = fg * c2[rt] + ig * gg, hn = og * tanhf_(cn);
        c2[rt] = cn; hb[rt] = bfbits(hn);
        if (row < B) {
          if (p.gates[1]) { float* gp = p.gates[1] + ((size_t)t * B + row) * 4 * HD + unit; gp[0] = ig; gp[HD] = fg; gp[2 * HD] = og; gp[3 * HD] = gg; }
          const size_t o = (size_t)(t + 1) * slot + (size_t)row * HD + unit;
          p.cs[1][o] = cn; p.hs[1][o] = hn; p.hsb[1][o] = (bf16_t)hn;
          const size_t oc = ((size_t)t * B + row) * 2 * HD + HD + unit;                      // JoinTable [c ; h_top], LSTM.lua:153
          p.cat[oc] = hn; p.cat_b[oc] = (bf16_t)hn;
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const unsigned h1v = __shfl(hb[rt], lane + 16, 64), h2v = __shfl(hb[rt], lane + 32, 64), h3v = __shfl(hb[rt], lane + 48, 64);
        if (q == 0) dst_granules(xg + (size_t)(2 * 2 + par) * 8192 + ((size_t)(member * 32 + 16 * rt + c16) * 8 + 2 * wave), u32x4{hb[rt] | (h1v << 16), tagc, h2v | (h3v << 16), tagc}, local);
      }
    }
    // =================== attention of row `member` of the group
    {
      const bool okg = gather<8, 256>(xg + (size_t)(2 * 2 + par) * 8192, tagc, H2, tid, p.err, 13);
      if (!okg) s_dead = 1;
      __syncthreads();
      if (s_dead) { dead = true; break; }
      // scores of this member's row against the pre-multiplied context: s[tt] = ctxA[row][tt] . h2[row]
      const int arow = row0 + member;                              // the batch row whose attention this workgroup computes
      const bool rvalid = arow < B;
      float h2v[8];
      {
        const bf16x8 hv = *reinterpret_cast<const bf16x8*>(H2 + (size_t)member * PA + lane * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) h2v[e] = (float)hv[e];
      }
      const bf16_t* ca = p.ctxa + ((size_t)min(arow, B - 1) * T) * HD + lane * 8;
      const bf16_t* cx = p.ctxb + ((size_t)min(arow, B - 1) * T) * HD + lane * 8;
      for (int tt = wave; tt < T; tt += 4) {
        const bf16x8 cv = *reinterpret_cast<const bf16x8*>(ca + (size_t)tt * HD);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf((float)cv[e], h2v[e], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) sc[tt] = s;
      }
      __syncthreads();
      if (wave == 0) {                                             // softmax over T (LSTM.lua:139)
        float m = -INFINITY;
        for (int tt = lane; tt < T; tt += 64) m = fmaxf(m, sc[tt]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float sum = 0.f;
        for (int tt = lane; tt < T; tt += 64) { const float e = expf(sc[tt] - m); sc[tt] = e; sum += e; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = 1.f / sum;
        for (int tt = lane; tt < T; tt += 64) { const float a = sc[tt] * inv; sc[tt] = a; if (rvalid) p.a_all[((size_t)t * B + arow) * T + tt] = a; }
      }
      __syncthreads();
      float cacc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cacc[e] = 0.f;
      for (int tt = wave; tt < T; tt += 4) {
        const bf16x8 cv = *reinterpret_cast<const bf16x8*>(cx + (size_t)tt * HD);
        const float a = sc[tt];
#pragma unroll
        for (int e = 0; e < 8; ++e) cacc[e] = fmaf(a, (float)cv[e], cacc[e]);
      }
      float* part = sc + 256;                                      // [4 waves][512]
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8) = f32x4{cacc[0], cacc[1], cacc[2], cacc[3]};
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8 + 4) = f32x4{cacc[4], cacc[5], cacc[6], cacc[7]};
      __syncthreads();
      if (tid < 128) {                                             // c[4 tid .. 4 tid + 3]: sum over the waves, publish + keep for the backward pass
        f32x4 v = *reinterpret_cast<const f32x4*>(part + tid * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(part + w * HD + tid * 4);
        dst_granules(xg + (size_t)(3 * 2 + par) * 8192 + ((size_t)member * 256 + 2 * tid), u32x4{bfbits(v[0]) | (bfbits(v[1]) << 16), tagc, bfbits(v[2]) | (bfbits(v[3]) << 16), tagc}, local);
        if (rvalid) {
          const size_t oc = ((size_t)t * B + arow) * 2 * HD + 4 * tid;
          *reinterpret_cast<f32x4*>(p.cat + oc) = v;
          *reinterpret_cast<u32x2*>(p.cat_b + oc) = u32x2{bfbits(v[0]) | (bfbits(v[1]) << 16), bfbits(v[2]) | (bfbits(v[3]) << 16)};
        }
      }
    }
    // =================== out = tanh(W_c [c ; h2]), LSTM.lua:153-157
    {
      const bool okg = gather<256, 8>(xg + (size_t)(3 * 2 + par) * 8192, tagc, F, tid, p.err, 14);
      if (!okg) s_dead = 1;
      __syncthreads();
      if (s_dead) { dead = true; break; }
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const unsigned char* src = wave < 2 ? F : H2;                // k = 256 wave + 32 s: waves 0, 1 read c, waves 2, 3 read h2
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt + c16) * PA + (256 * (wave & 1) + 32 * s + 8 * q) * 2);
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcr[s], bv, acc[rt], 0, 0, 0);
        }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) *reinterpret_cast<f32x4*>(red + ((wave * 2 + rt) * 64 + lane) * 4) = acc[rt];
      __syncthreads();
      if (wave < 2) {
        const int rt = wave;
        f32x4 v = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + rt) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((w * 2 + rt) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tanhf_(v[i]);
        const int row = row0 + 16 * rt + c16;
        if (t + 1 < L) dst_granules(xg + (size_t)(0 * 2 + par) * 8192 + ((size_t)(member * 32 + 16 * rt + c16) * 8 + 2 * q), u32x4{bfbits(v[0]) | (bfbits(v[1]) << 16), tagc, bfbits(v[2]) | (bfbits(v[3]) << 16), tagc}, local);
        if (row < B) {
          const size_t o = (size_t)(t + 1) * slot + (size_t)row * HD + 16 * member + 4 * q;
          *reinterpret_cast<f32x4*>(p.out + o) = v;
          *reinterpret_cast<u32x2*>(p.out_b + o) = u32x2{bfbits(v[0]) | (bfbits(v[1]) << 16), bfbits(v[2]) | (bfbits(v[3]) << 16)};
        }
      }
      if (t + 1 < L) dma_zx(t + 1);
      __syncthreads();                                             // red / sc are reused by the next step
    }
  }
  dwait0();
}

// ---------------------------------------------------------------------------------------------
size_t dec_cluster_xbuf_bytes(int B) { return (size_t)((B + R - 1) / R) * 4 * 2 * 8192 * sizeof(u64); }
size_t dec_cluster_xtab_bytes(int B) { return (size_t)((B + R - 1) / R) * NM * sizeof(u64) + 256; }
bool dec_cluster_supported(int Hd, int Ld, int input_feed, int T, int L, int cus) { return Hd == HD && Ld == 2 && input_feed && T >= 1 && T <= 256 && L + 2 < 4096 && cus >= 8 * NM; }

void dec_cluster_forward(hipStream_t s, const DecClFwdArgs& a0) {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  const size_t lds = (size_t)3 * R * PA + 8192 + 8192 + 1024 + 8192;
  (void)hipFuncSetAttribute((const void*)dec_cl_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClFwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    hipLaunchKernelGGL(dec_cl_fwd_kernel, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), lds, s, a);
  }
}

}  // namespace aocr
