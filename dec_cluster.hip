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
constexpr int LDS_BYTES = 3 * R * PA + 32768;                    // three operand buffers + the reduction scratch

// ---- VMEM in program order: polls first, the previous phase's output stores behind them, then `s_waitcnt vmcnt(#stores)` -- the
// polls are waited for, the stores are not (vmcnt counts loads and stores in issue order on gfx9).  Every store below is ONE
// instruction whatever the lane's predicate (invalid lanes point at a trash slot), so the counts are exact.
// L1-bypassing loads (sc1): scalar base + 32-bit lane offset
// (group inside one XCD: its L2 is the point of coherence; otherwise system scope on both sides)
__device__ __forceinline__ void ld16_sc1(u32x4& v, unsigned voff, const void* sbase, bool local) {
  if (local) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void ld4_sc1(unsigned& v, unsigned voff, const void* sbase, bool local) {
  if (local) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
}
// exchange payload / flag stores
__device__ __forceinline__ void pst4(void* p, unsigned v, bool local) {
  if (local) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void pst8(void* p, u32x2 v, bool local) {
  if (local) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dpin(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void st4(void* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4f(void* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8(void* p, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st16f(void* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
// workgroup barrier that orders LDS only: global loads / stores stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ unsigned bfbits(float x) { bf16_t h = (bf16_t)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
__device__ __forceinline__ unsigned bfpair(float a, float b) { return bfbits(a) | (bfbits(b) << 16); }
__device__ __forceinline__ u64 ldg64(const u64* p) { return __hip_atomic_load(const_cast<u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-wide reductions on the DPP network (no LDS round trips): butterflies inside a row of 16, then row_bcast15 / row_bcast31
template <int CTRL, int ROWS> __device__ __forceinline__ float dppf(float ident, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
template <class OP> __device__ __forceinline__ float wave_reduce(float v, float ident, OP op) {
  v = op(v, dppf<0xB1, 0xF>(ident, v));          // quad_perm [1,0,3,2]
  v = op(v, dppf<0x4E, 0xF>(ident, v));          // quad_perm [2,3,0,1]
  v = op(v, dppf<0x141, 0xF>(ident, v));         // row_half_mirror
  v = op(v, dppf<0x140, 0xF>(ident, v));         // row_mirror: every lane of a row holds the row's result
  v = op(v, dppf<0x142, 0xA>(ident, v));         // row_bcast15 into rows 1, 3
  v = op(v, dppf<0x143, 0xC>(ident, v));         // row_bcast31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// All-gather of one 32 x 512 bf16 operand into an LDS buffer.  The payload is the real output tensor (h, [c ; h] or out as bf16): every
// wave of every member stores its piece, waits for the write acknowledgement, then raises its flag (128 flags of 4 bytes per
// operand); a reader polls the flags -- 512 bytes per round instead of the operand -- and loads the 32 KB once.
//   payload():  this wave's piece (any number of stores);   deferred(): exactly NST one-instruction stores of older outputs, issued
//   while the piece's acknowledgement is awaited;   pre(): loads the caller wants in flight beside the polls.
template <int NST, class PAY, class DEF, class PRE>
__device__ __forceinline__ void gather(unsigned* flags, unsigned tag, const bf16_t* src, int stride_bytes, int row0, int B, unsigned char* dst, int tid, int wave,
                                       int member, bool local, int* err, int code, int* dead_flag, PAY&& payload, DEF&& deferred, PRE&& pre, u64 (&gs)[3], bool stamps) {
  u64 t0 = stamps ? __builtin_readcyclecounter() : 0;
  payload();
  deferred();
  wait_vm<NST>();                               // the piece is in L2
  if ((tid & 63) == 0) pst4(flags + member * 4 + wave, tag, local);
  pre();
  if (stamps) { const u64 t1 = __builtin_readcyclecounter(); gs[0] += t1 - t0; t0 = t1; }
  const unsigned foff = (unsigned)(tid & 63) * 4;
  unsigned f0, f1; int spins = 0;
#pragma nounroll
  while (true) {
    ld4_sc1(f0, foff, flags, local); ld4_sc1(f1, foff + 256, flags, local);
    wait_vm<0>();
    asm volatile("" : "+v"(f0), "+v"(f1));
    if (__all(f0 == tag && f1 == tag)) break;
    asm volatile("" : "+s"(spins));             // keeps the compiler from reasoning about (and unrolling over) the trip count
    if (++spins > DC_SPIN_LIMIT) { if ((tid & 63) == 0) { atomicExch(err, code); *dead_flag = 1; } break; }
    __builtin_amdgcn_s_sleep(1);
  }
  if (stamps) { const u64 t1 = __builtin_readcyclecounter(); gs[1] += t1 - t0; t0 = t1; }
  u32x4 gr[8];                                  // 32 rows x 64 chunks of 16 bytes: thread -> chunk tid & 63 of rows (tid >> 6) + 4 j
#pragma unroll
  for (int j = 0; j < 8; ++j) ld16_sc1(gr[j], (unsigned)(min(row0 + (tid >> 6) + 4 * j, B - 1) * stride_bytes + (tid & 63) * 16), src, local);
  lds_barrier();                                // every wave of this workgroup is past its reads of the previous contents of dst
  wait_vm<0>();
#pragma unroll
  for (int j = 0; j < 8; ++j) dpin(gr[j]);
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(dst + (size_t)((tid >> 6) + 4 * j) * PA + (tid & 63) * 16) = gr[j];
  lds_barrier();
  if (stamps) { const u64 t1 = __builtin_readcyclecounter(); gs[2] += t1 - t0; }
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

#define DC_STAMP(k) do { if (p.stamps) { const u64 now_ = __builtin_readcyclecounter(); stamp[k] += now_ - tprev; tprev = now_; } } while (0)

__global__ __launch_bounds__(256, 1) void dec_cl_fwd_kernel(DecClFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const F = lds;                        // feed (out(t-1)), later c(t)
  unsigned char* const H1 = lds + R * PA;              // h1(t-1), later h1(t)
  unsigned char* const H2 = lds + 2 * R * PA;          // h2(t-1), later h2(t)
  float* const red = reinterpret_cast<float*>(lds + 3 * R * PA);     // [4 waves][4 tiles][2][64 lanes][4]: K-split partial tiles (32 KB);
  float* const part = red;                                            // attention: [4 waves][512] partial context,
  float* const sc = red + 2048;                                       //            [256] scores
  __shared__ int s_local, s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % NM, gl = (i8 / NM) * 8 + xcd;
  if (gl >= p.ngroups) return;
  const int group = p.group0 + gl;
  const int B = p.B, T = p.T, L = p.L, row0 = group * R;
  const int unit = 16 * member + 4 * wave + q;                   // gate epilogues: this lane's hidden unit (lane = (batch row c16, unit))
  const size_t slot = (size_t)B * HD;

  // ---- co-location check (rnn_cluster.hip): plain granule stores are only visible to the group's polls inside one XCD
  u64* const xt = p.xtab + (size_t)group * NM;
  if (tid == 0) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u;
    stg64(xt + member, ((u64)p.epoch << 32) | (u64)(xcc + 1u));
    int same = 1;
    for (int m = 0; m < NM; ++m) {
      u64 v; int spins = 0;
      while ((unsigned)((v = ldg64(xt + m)) >> 32) != p.epoch) { if (++spins > DC_SPIN_LIMIT) { atomicExch(p.err, 15); same = 0; break; } __builtin_amdgcn_s_sleep(2); }
      if ((unsigned)v != xcc + 1u) same = 0;
    }
    s_local = same && !p.force_remote; s_dead = 0;
  }
  __syncthreads();
  const bool local = __builtin_amdgcn_readfirstlane(s_local) != 0;

  // ---- resident weights: MFMA A fragments.  The four waves split K (wave w: columns 256 w .. 256 w + 255 of the concatenated
  // operand) and each holds all 64 gate rows of the member (4 tiles of [unit][gate] rows), so an operand fragment read from LDS
  // feeds four products; the partial tiles meet in LDS, wave w finishes tile w (units 4w .. 4w+3).
  bf16x8 w1a[16], w1b[16], w2a[16], w2b[16], wcr[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) wcr[s] = *reinterpret_cast<const bf16x8*>(p.wc + (size_t)(16 * member + c16) * 2 * HD + 256 * wave + 32 * s + 8 * q);
  const size_t xwf = (p.exp & 2) ? 0 : 1;
  auto wfrag = [&](const bf16_t* wi, const bf16_t* wh, int c16_, int q_, int j, int s) {
    return *reinterpret_cast<const bf16x8*>((wave < 2 ? wi : wh) + xwf * ((size_t)((c16_ & 3) * HD + 16 * member + 4 * j + (c16_ >> 2)) * HD + 256 * (wave & 1) + 32 * s + 8 * q_));
  };
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      w1a[j * 8 + s] = wfrag(p.w1i, p.w1h, c16, q, j, s); w1b[j * 8 + s] = wfrag(p.w1i, p.w1h, c16, q, j + 2, s);
      w2a[j * 8 + s] = wfrag(p.w2i, p.w2h, c16, q, j, s); w2b[j * 8 + s] = wfrag(p.w2i, p.w2h, c16, q, j + 2, s);
    }
  float b2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b2[i] = p.b2i[i * HD + unit] + p.b2h[i * HD + unit];
  float c1[2], c2[2], zxr[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = min(row0 + 16 * rt + c16, B - 1);
    c1[rt] = p.cs[0][(size_t)row * HD + unit]; c2[rt] = p.cs[1][(size_t)row * HD + unit];
#pragma unroll
    for (int i = 0; i < 4; ++i) zxr[rt][i] = p.zx1[(size_t)row * 4 * HD + i * HD + unit];
  }
  unsigned* const xg = reinterpret_cast<unsigned*>(p.xbuf) + (size_t)group * 4 * 128;      // flags [kind: out, h1, h2, c][member][wave]
  load_rows(p.out_b, row0, B, F, tid); load_rows(p.hsb[0], row0, B, H1, tid); load_rows(p.hsb[1], row0, B, H2, tid);
  __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): nothing of the prologue is in flight inside the loop
  __syncthreads();
  u64 stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, gs[3] = {0, 0, 0}, tprev = __builtin_readcyclecounter();

  // the row of the group whose attention this workgroup computes
  const int arow = row0 + member; const bool rvalid = arow < B;
  const bf16_t* const ca = p.ctxa + (size_t)min(arow, B - 1) * T * HD;
  const bf16_t* const cx = p.ctxb + (size_t)min(arow, B - 1) * T * HD;
  const size_t xa = (p.exp & 1) ? 0 : 1, xw = (p.exp & 2) ? 0 : 1;      // perf experiments (AOCR_DC_EXP): collapse the streamed addresses
  const int ntile = (T + 15) >> 4;

  // one K-split product: acc tiles -> LDS -> this wave's tile (both row tiles) summed over the four waves
  auto product = [&](const bf16x8 (&wa)[16], const bf16x8 (&wb)[16], const unsigned char* x0, const unsigned char* x1, f32x4 (&v)[2]) {
    const unsigned char* src = (wave < 2 ? x0 : x1) + (256 * (wave & 1) + 8 * q) * 2;
    f32x4 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[j][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt + c16) * PA + 64 * s);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? wa[j * 8 + s] : wb[(j - 2) * 8 + s], bv, acc[j][rt], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) *reinterpret_cast<f32x4*>(red + ((size_t)((wave * 4 + j) * 2 + rt) * 64 + lane) * 4) = acc[j][rt];
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      v[rt] = *reinterpret_cast<const f32x4*>(red + ((size_t)((0 * 4 + wave) * 2 + rt) * 64 + lane) * 4);
#pragma unroll
      for (int w2_ = 1; w2_ < 4; ++w2_) v[rt] += *reinterpret_cast<const f32x4*>(red + ((size_t)((w2_ * 4 + wave) * 2 + rt) * 64 + lane) * 4);
    }
  };
  // LSTM cell on this lane's (unit, row) pairs; returns the packed h of the four units of (row, wave) in lanes q == 0
  auto cell = [&](const f32x4 (&z)[2], float (&c)[2], f32x4 (&g)[2], u32x2 (&hp)[2]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float ig = sigmoidf_(z[rt][0]), fg = sigmoidf_(z[rt][1]), og = sigmoidf_(z[rt][2]), gg = tanhf_(z[rt][3]);
      const float cn = fg * c[rt] + ig * gg, hn = og * tanhf_(cn);
      c[rt] = cn; g[rt] = f32x4{ig, fg, og, gg};
      const unsigned hb = bfbits(hn);
      const unsigned h1v = __shfl(hb, lane + 16, 64), h2v = __shfl(hb, lane + 32, 64), h3v = __shfl(hb, lane + 48, 64);
      hp[rt] = u32x2{hb | (h1v << 16), h2v | (h3v << 16)};
    }
  };
  // deferred stores of one LSTM layer (4 instructions; 6 with the [c ; h] shadow of the top layer); the bf16 h is the exchange payload
  // (ot: the thread id through an opaque per-step copy, so that the address arithmetic stays inside the step instead of being hoisted
  // out of the loop into ~200 live registers)
  auto store_layer = [&](int ot, int l, int t, const f32x4 (&g)[2], const float (&c)[2], const u32x2 (&hp)[2], bool top) {
    const int c16 = ot & 15, q = (ot >> 4) & 3, unit = 16 * member + 4 * wave + q;
    unsigned char* const trash = reinterpret_cast<unsigned char*>(p.err + 16) + ot * 16;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int row = row0 + 16 * rt + c16; const bool ok = row < B;
      st16f(ok && p.gates[l] ? (void*)(p.gates[l] + (((size_t)t * B + row) * HD + unit) * 4) : (void*)trash, g[rt]);
      st4f(ok ? (void*)(p.cs[l] + (size_t)(t + 1) * slot + (size_t)row * HD + unit) : (void*)trash, c[rt]);
      const bool okh = ok && q == 0;
      if (top) st8(okh ? (void*)(p.cat_b + ((size_t)t * B + row) * 2 * HD + HD + 16 * member + 4 * wave) : (void*)trash, hp[rt]);   // JoinTable [c ; h_top], LSTM.lua:153
    }
  };
  f32x4 ov = f32x4{0.f, 0.f, 0.f, 0.f}; u32x2 ovb = u32x2{0u, 0u};       // out(t) of this lane (waves 0, 1), stored one phase late
  auto store_out = [&](int ot, int t) {                                  // 1 instruction (the bf16 copy is the exchange payload)
    const int c16 = ot & 15, q = (ot >> 4) & 3;
    unsigned char* const trash = reinterpret_cast<unsigned char*>(p.err + 16) + ot * 16;
    const int row = row0 + 16 * wave + c16; const bool ok = wave < 2 && row < B;
    const size_t o = (size_t)(t + 1) * slot + (size_t)row * HD + 16 * member + 4 * q;
    st16f(ok ? (void*)(p.out + o) : (void*)trash, ov);
  };

  for (int t = 0; t < L; ++t) {
    const unsigned tagc = p.epoch * 4096u + (unsigned)(t + 1);    // flags of step t: + kind * 1024

    int ot = tid; asm volatile("" : "+v"(ot));                     // opaque copy of the thread id (see store_layer)
    const int olane = ot & 63, oc16 = ot & 15, oq = (ot >> 4) & 3, ounit = 16 * member + 4 * wave + oq;
    unsigned char* const otrash = reinterpret_cast<unsigned char*>(p.err + 16) + ot * 16;
    // =================== layer 1: z1 = [feed ; h1(t-1)] W1^T + zx1(t)
    if (t > 0) {
      gather<1>(xg + 0 * 128, tagc - 1u, p.out_b + (size_t)t * slot, HD * 2, row0, B, F, ot, wave, member, local, p.err, 11, &s_dead,
                [&] {                                               // out(t-1) of this member's 16 units (waves 0, 1: one row tile each)
                  const int row = row0 + 16 * wave + oc16;
                  pst8(wave < 2 && row < B ? (void*)(p.out_b + (size_t)t * slot + (size_t)row * HD + 16 * member + 4 * oq) : (void*)otrash, ovb, local);
                },
                [&] { store_out(ot, t - 1); }, [] {}, gs, p.stamps != nullptr);
      if (s_dead) break;
    }
    DC_STAMP(0);
    f32x4 g1[2]; u32x2 hp1[2];
    {
      f32x4 z[2];
      product(w1a, w1b, F, H1, z);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[rt][i] += zxr[rt][i];
      cell(z, c1, g1, hp1);
    }
    DC_STAMP(1);
    // =================== layer 2: z2 = [h1(t) ; h2(t-1)] W2^T + b
    auto publish_h = [&](int l, const u32x2 (&hp)[2]) {          // h(t) of (row, wave): lanes q == 0 hold the four units packed
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int row = row0 + 16 * rt + oc16;
        pst8(oq == 0 && row < B ? (void*)(p.hsb[l] + (size_t)(t + 1) * slot + (size_t)row * HD + 16 * member + 4 * wave) : (void*)otrash, hp[rt], local);
      }
    };
    gather<4>(xg + 1 * 128, tagc + 1024u, p.hsb[0] + (size_t)(t + 1) * slot, HD * 2, row0, B, H1, ot, wave, member, local, p.err, 12, &s_dead,
              [&] { publish_h(0, hp1); }, [&] { store_layer(ot, 0, t, g1, c1, hp1, false); },
              [] {}, gs, p.stamps != nullptr);
    if (s_dead) break;
    DC_STAMP(2);
    f32x4 g2[2]; u32x2 hp2[2];
    {
      f32x4 z[2];
      product(w2a, w2b, H1, H2, z);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[rt][i] += b2[i];
      cell(z, c2, g2, hp2);
    }
    DC_STAMP(3);
    // =================== attention of row `member` of the group: scores on MFMA (A = a 16-step tile of ctx . W_a, B = h2 broadcast)
    float av; unsigned cb;
    {
      bf16x8 cav[16];                                              // tile `wave` of the pre-multiplied context: in flight beside the polls
      const bf16_t* carow = ca + xa * ((size_t)min(16 * wave + oc16, T - 1) * HD + 8 * oq);
      gather<6>(xg + 2 * 128, tagc + 2048u, p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, row0, B, H2, ot, wave, member, local, p.err, 13, &s_dead,
                [&] { publish_h(1, hp2); }, [&] { store_layer(ot, 1, t, g2, c2, hp2, true); },
                [&] {
#pragma unroll
                  for (int s = 0; s < 16; ++s) cav[s] = *reinterpret_cast<const bf16x8*>(carow + xa * 32 * s);
                }, gs, p.stamps != nullptr);
      if (s_dead) break;
      DC_STAMP(4);
      const unsigned char* hrow = H2 + (size_t)member * PA + 16 * q;
      for (int tile = wave; tile < ntile; tile += 4) {
        if (tile != wave) {
          const bf16_t* r2 = ca + (size_t)min(16 * tile + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
          for (int s = 0; s < 16; ++s) cav[s] = *reinterpret_cast<const bf16x8*>(r2 + 32 * s);
        }
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cav[s], *reinterpret_cast<const bf16x8*>(hrow + 64 * s), acc, 0, 0, 0);
        if (c16 == 0) *reinterpret_cast<f32x4*>(sc + 16 * tile + 4 * q) = acc;       // rows 4q .. 4q+3 of the tile; every column holds the same value
      }
      // context rows of this wave: tt = 64 c + 4 i + wave; the first chunk is loaded while the softmax runs
      bf16x8 cv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + xa * ((size_t)min(4 * i + wave, T - 1) * HD + 8 * olane));
      lds_barrier();
      DC_STAMP(10);
      // softmax over T (LSTM.lua:139), redundantly in every wave: lane holds steps lane + 64 j
      float aj[4];
      {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) { aj[j] = lane + 64 * j < T ? sc[lane + 64 * j] : -INFINITY; m = fmaxf(m, aj[j]); }
        m = wave_reduce(m, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { aj[j] = lane + 64 * j < T ? expf(aj[j] - m) : 0.f; sum += aj[j]; }
        sum = wave_reduce(sum, 0.f, [](float a, float b) { return a + b; });
        const float inv = 1.f / sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) aj[j] *= inv;
      }
      av = wave == 0 ? aj[0] : wave == 1 ? aj[1] : wave == 2 ? aj[2] : aj[3];      // a[tid]
      DC_STAMP(11);
      float cacc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cacc[e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (64 * c >= T) break;
        if (c > 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + (size_t)min(64 * c + 4 * i + wave, T - 1) * HD + 8 * olane);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, aj[c]), 4 * i + wave));     // 0 beyond T
#pragma unroll
          for (int e = 0; e < 8; ++e) cacc[e] = fmaf(a, (float)cv[i][e], cacc[e]);
        }
      }
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8) = f32x4{cacc[0], cacc[1], cacc[2], cacc[3]};
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8 + 4) = f32x4{cacc[4], cacc[5], cacc[6], cacc[7]};
      lds_barrier();
      DC_STAMP(12);
      {                                                            // c[2 tid], c[2 tid + 1]: sum over the waves, publish one granule
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { v0 += part[w * HD + 2 * tid]; v1 += part[w * HD + 2 * tid + 1]; }
        cb = bfpair(v0, v1);
      }
    }
    DC_STAMP(5);
    // =================== out = tanh(W_c [c ; h2]), LSTM.lua:153-157
    {
      const int tn = min(t + 1, L - 1);
      gather<1>(xg + 3 * 128, tagc + 3072u, p.cat_b + (size_t)t * B * 2 * HD, HD * 4, row0, B, F, ot, wave, member, local, p.err, 14, &s_dead,
                [&] { pst4(rvalid ? (void*)(p.cat_b + ((size_t)t * B + arow) * 2 * HD + 2 * ot) : (void*)otrash, cb, local); },      // c of row `member`: units 2 tid, 2 tid + 1
                [&] { st4f(rvalid && ot < T ? (void*)(p.a_all + ((size_t)t * B + arow) * T + ot) : (void*)otrash, av); },
                [&] {                                              // zx1 of the next step: landed with the polls
#pragma unroll
                  for (int rt = 0; rt < 2; ++rt) {
                    const int row = min(row0 + 16 * rt + oc16, B - 1);
#pragma unroll
                    for (int i = 0; i < 4; ++i) zxr[rt][i] = p.zx1[((size_t)tn * B + row) * 4 * HD + i * HD + ounit];
                  }
                }, gs, p.stamps != nullptr);
      if (s_dead) break;
      DC_STAMP(6);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const unsigned char* src = (wave < 2 ? F : H2) + (256 * (wave & 1) + 8 * q) * 2;   // k = 256 wave + 32 s: waves 0, 1 read c, waves 2, 3 read h2
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcr[s], *reinterpret_cast<const bf16x8*>(src + (size_t)(16 * rt + c16) * PA + 64 * s), acc[rt], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) *reinterpret_cast<f32x4*>(red + ((wave * 2 + rt) * 64 + lane) * 4) = acc[rt];
      lds_barrier();
      if (wave < 2) {                                              // wave = row tile
        f32x4 v = *reinterpret_cast<const f32x4*>(red + ((0 * 2 + wave) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((w * 2 + wave) * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tanhf_(v[i]);
        ov = v; ovb = u32x2{bfpair(v[0], v[1]), bfpair(v[2], v[3])};         // published behind the polls of the next step's first gather
      }
    }
    DC_STAMP(7);
  }
  if (!s_dead) {
    store_out(tid, L - 1);
    const int row = row0 + 16 * wave + c16;
    if (wave < 2 && row < B) *reinterpret_cast<u32x2*>(p.out_b + (size_t)L * slot + (size_t)row * HD + 16 * member + 4 * q) = ovb;
  }
  wait_vm<0>();
  if (p.stamps && wid == 0 && tid == 0)
    { stamp[8] = gs[0]; stamp[9] = gs[1]; stamp[13] = gs[2]; stamp[15] = local ? 1 : 0; for (int k = 0; k < 16; ++k) p.stamps[k] = stamp[k]; }
}

// ---------------------------------------------------------------------------------------------
size_t dec_cluster_xbuf_bytes(int B) { return (size_t)((B + R - 1) / R) * 4 * 128 * sizeof(unsigned) + 256; }
size_t dec_cluster_xtab_bytes(int B) { return (size_t)((B + R - 1) / R) * NM * sizeof(u64) + 256; }
bool dec_cluster_supported(int Hd, int Ld, int input_feed, int T, int L, int cus) { return Hd == HD && Ld == 2 && input_feed && T >= 1 && T <= 256 && L + 2 < 1024 && cus >= 8 * NM; }

void dec_cluster_forward(hipStream_t s, const DecClFwdArgs& a0) {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  const size_t lds = LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)dec_cl_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClFwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    a.exp = getenv("AOCR_DC_EXP") ? atoi(getenv("AOCR_DC_EXP")) : 0;
    a.stamps = getenv("AOCR_DC_STAMPS") ? a.xtab + (size_t)groups * NM : nullptr;   // debugging aid: cycles per phase of workgroup 0
    hipLaunchKernelGGL(dec_cl_fwd_kernel, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), lds, s, a);
  }
}

}  // namespace aocr
