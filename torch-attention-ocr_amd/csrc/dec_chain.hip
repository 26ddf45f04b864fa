// dec_chain.hip -- the whole-sequence decoder kernels of dec_cluster.hip (teacher-forced loop model.lua:553-568 / gold pass :604-627, its
// BPTT model.lua:643-661; cell LSTM.lua:18-122, attention LSTM.lua:124-162) rebuilt around what bounds them: a step of the loop is a chain
// of four (forward) / five (backward) all-gathers inside a group of 32 compute units, and in dec_cluster.hip a step costs 18.8 us / 25 us
// of which the MFMAs are ~1.2 us -- the rest is exchange latency (store -> acknowledgement -> flag -> poll -> load) and the latency of
// the short dependent phases between them.  Two changes:
//  (1) TWO CHAINS.  The 32 batch rows of a group are two independent chains of 16 rows (the recurrence never mixes batch rows).  Every
//      phase of a step runs for chain 0, then for chain 1, so a chain's exchange is in flight while the other chain computes; and the
//      attention of chain 0's rows (members 0-15) runs at the same time as that of chain 1's rows (members 16-31).
//  (2) TAG-FREE EXCHANGE.  The payload is still the real output tensor (h, [c ; h], out, d pre, d q, d z as bf16; d c as fp32), but there is
//      no acknowledgement wait and no flag: the destination is pre-filled with a bit pattern no payload can carry (0xFFFFFFFF per dword: two
//      bf16 NaNs with every mantissa bit set / an fp32 NaN the arithmetic never produces; a payload dword that came out as this pattern
//      is stored as 0xFFFEFFFF, the same NaNs), and a reader simply loads the operand and looks for dwords that are still the pattern
//      (a dword store is atomic; the readers' loads bypass L1).  The loads of the operand of phase n+1 are issued at the START of phase n
//      (its producers published one phase earlier) and consumed at the start of phase n+1: the common case costs no exposed round trip.
//      Pre-fill: every member fills its OWN pieces -- steps 0 and 1 before the group's co-location handshake (write-through, acknowledged),
//      step t+2 at the end of step t (acknowledged long before the member publishes the last payload of step t+1, which is what every
//      reader of step t+2 has to see first).
// Layouts, weight residency (288 / 304 VGPRs of A fragments per lane), K split over the four waves, reduction order and every arithmetic
// operation are dec_cluster.hip's: the outputs are bit-identical to its kernels (tests/test_step_gpu.py compares all three paths).
#include "ops.h"
#include "dec_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace aocr {
namespace {
constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr int NCH = 2, RC = 16;                                     // chains per group, rows per chain
constexpr int OPB = RC * PA;                                        // one operand buffer of a chain: 16 rows x 1 KB (+ pad)
constexpr int CH_FWD_LDS = NCH * 3 * OPB + 16384 + 8192 + 1024 + NCH * 5 * 1024;     // operands + K-split partial tiles + attention partial context + scores + the next step's gate inputs / tokens

__device__ __forceinline__ unsigned sane(unsigned x) { return x == SENT ? 0xFFFEFFFFu : x; }
__device__ __forceinline__ unsigned umax4(const u32x4& v) { return max(max(v[0], v[1]), max(v[2], v[3])); }
__device__ __forceinline__ void ld4g(unsigned& v, const void* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void dpin1(unsigned& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pst16(void* p, u32x4 v, bool local) {
  if (local) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

#ifdef DC_DEBUG_STAMPS
#define CH_STAMP(k) do { if (p.stamps) { const u64 now_ = __builtin_readcyclecounter(); stamp[k] += now_ - tprev; tprev = now_; } } while (0)
#else
#define CH_STAMP(k) do { } while (0)
#endif

// LDS-DMA (global_load_lds): the load writes LDS directly -- 64 lanes x 16 (4) bytes lane-linearly at the wave-uniform byte address in M0 -- and
// has no VGPR destination.  That is what lets an operand be in flight ACROSS a whole compute phase: the destination of an asm load into
// registers is a value hipcc may copy (spill to an AGPR, re-allocate) before the data has arrived.  Inline asm, so the loads are outside
// hipcc's vmcnt bookkeeping (it would drain them at the next barrier); M0 is written in the statement that reads it (cdna_hip_programming.md).
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void dma16x(const void* g, unsigned lds_dst, bool local) {
  unsigned keep;
  if (local) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
  else asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc0 sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* g, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// The operand of one chain: 16 rows x 1 KB, one row per wave instruction; wave w fetches rows (w + 4 j + member) & 15 (rotated by the member
// index so that the members of a group start at different rows) into dst (row pitch `pitch`, byte offset chunk0 inside a row).
__device__ __forceinline__ void pend_issue(const void* src, int stride_bytes, int rbase, int B, unsigned char* dst, int pitch, int lane, int wave, int member, bool local, int chunk0 = 0) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rl = (wave + 4 * j + member) & (RC - 1);
    dma16x(reinterpret_cast<const unsigned char*>(src) + (size_t)min(rbase + rl, B - 1) * stride_bytes + chunk0 + lane * 16,
           __builtin_amdgcn_readfirstlane(lds_addr(dst) + rl * pitch + chunk0), local);
  }
}
// Wait for the loads (NST = the one-instruction stores issued after them: those are not waited for), look -- every thread at the 4 x 16 bytes its
// lane fetched -- for dwords nobody has written yet, fetch the rows that have some again until none is left; then the workgroup barrier.
template <int NST>
__device__ __forceinline__ void pend_land(const void* src, int stride_bytes, int rbase, int B, unsigned char* dst, int pitch, int lane, int wave, int member, bool local,
                                          int* err, int code, int* dead_flag, int chunk0 = 0) {
  wait_vm<NST>();
  u32x4 g[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = *reinterpret_cast<const u32x4*>(dst + (size_t)((wave + 4 * j + member) & (RC - 1)) * pitch + chunk0 + lane * 16);
  unsigned mx = max(max(umax4(g[0]), umax4(g[1])), max(umax4(g[2]), umax4(g[3])));
  if (__any(mx == SENT)) {
    int spins = 0;
#pragma nounroll
    while (true) {
      asm volatile("" : "+v"(spins));
      if (++spins > DC_SPIN_LIMIT) { if (lane == 0) { atomicExch(err, code); *dead_flag = 1; } break; }
      __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (__any(umax4(g[j]) == SENT)) {
          const int rl = (wave + 4 * j + member) & (RC - 1);
          dma16x(reinterpret_cast<const unsigned char*>(src) + (size_t)min(rbase + rl, B - 1) * stride_bytes + chunk0 + lane * 16,
                 __builtin_amdgcn_readfirstlane(lds_addr(dst) + rl * pitch + chunk0), local);
        }
      wait_vm<0>();
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = *reinterpret_cast<const u32x4*>(dst + (size_t)((wave + 4 * j + member) & (RC - 1)) * pitch + chunk0 + lane * 16);
      mx = max(max(umax4(g[0]), umax4(g[1])), max(umax4(g[2]), umax4(g[3])));
      if (!__any(mx == SENT)) break;
    }
  }
  lds_barrier();
}
// step 0: the initial states from plain bf16 arrays [B][512]
__device__ __forceinline__ void load_rows16(const bf16_t* src, int rbase, int B, unsigned char* dst, int tid) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = tid + 256 * j, row = idx >> 6, ch = idx & 63;
    *reinterpret_cast<u32x4*>(dst + (size_t)row * PA + ch * 16) = *reinterpret_cast<const u32x4*>(src + (size_t)min(rbase + row, B - 1) * HD + ch * 8);
  }
}
template <int C> struct IC { static constexpr int value = C; };
}  // namespace

// =============================================================================================================================
// Forward: teacher-forced loop (train step, gold pass).  Phases of step t, each for chain 0 then chain 1 (operand <- published in):
//   P1  z1 = [out(t-1) ; h1(t-1)] W1^T + zx1(t) -> c1, h1(t)          out(t-1) <- P4 of step t-1        publishes h1(t)
//   P2  z2 = [h1(t) ; h2(t-1)] W2^T + b         -> c2, h2(t)          h1(t)    <- P1                    publishes h2(t)
//   P3  attention of row r on member r (members 16 c .. 16 c + 15 work for chain c, the others pass)     h2(t) <- P2      publishes c(t) of the row
//   P4  out(t) = tanh(W_c [c(t) ; h2(t)])                              c(t)     <- P3                    publishes out(t)
template <bool DEC, bool RES>     // RES: T <= 64 -- the 16-step tile of ctx . W_a a wave multiplies stays in its registers for the whole loop (64 VGPRs)
__global__ __launch_bounds__(256, 1) void dec_ch_fwd_kernel(DecClFwdArgs p) {
  static_assert(!DEC, "the greedy variant is not on the two-chain kernel yet");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* const red = reinterpret_cast<float*>(lds + NCH * 3 * OPB);            // [4 waves][4 tiles][64 lanes][4]: K-split partial tiles of one chain
  float* const part = reinterpret_cast<float*>(lds + NCH * 3 * OPB + 16384);   // attention: [4 waves][512] partial context
  float* const sc = part + 4 * HD;                                             //            [256] scores
  float* const zxs = sc + 256;                                                 // [chain][4 gates + token][256 threads]: zx1 of the next step, the token after it (LDS-DMA)
  __shared__ int s_local, s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % NM, gl = (i8 / NM) * 8 + xcd;
  if (gl >= p.ngroups) return;
  const int group = p.group0 + gl;
  const int B = p.B, T = p.T, L = p.L, row0 = group * R;
  const int unit = 16 * member + 4 * wave + q;
  const size_t slot = (size_t)B * HD;
  const int arow = row0 + member; const bool rvalid = arow < B;     // the row whose attention this workgroup computes
  const int mych = member >> 4;                                     // ... and its chain
  unsigned char* const trash0 = reinterpret_cast<unsigned char*>(p.err + 16);

  // ---- pre-fill of this member's pieces of step s: wave 0 out(s) [slot s + 1], wave 1 / 2 h1(s) / h2(s) [slot s + 1], wave 3 the c half of
  // [c ; h2](s) of the member's row.  ONE store per wave (invalid -> trash slot).
  auto prefill = [&](int s, int ot, bool loc) {
    const int ln = ot & 63; unsigned char* const trash = trash0 + ot * 16;
    void* dst;
    if (wave < 3) {
      bf16_t* const base = wave == 0 ? p.out_b : p.hsb[wave - 1];
      const int row = row0 + (ln >> 1);
      dst = (s < L && row < B) ? (void*)(base + (size_t)(s + 1) * slot + (size_t)row * HD + 16 * member + 8 * (ln & 1)) : (void*)trash;
    } else dst = (s < L && rvalid) ? (void*)(p.cat_b + ((size_t)s * B + arow) * 2 * HD + 8 * ln) : (void*)trash;
    pst16(dst, u32x4{SENT, SENT, SENT, SENT}, loc);
  };
  prefill(0, tid, false); prefill(1, tid, false);
  wait_vm<0>();
  __syncthreads();
  // ---- co-location check (rnn_cluster.hip); it is also the point after which every member's pre-fill of steps 0 and 1 is in memory
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

  // ---- resident weights (dec_cluster.hip): MFMA A fragments, wave w = columns 256 w .. of the concatenated operand, all 64 gate rows
  bf16x8 w1a[16], w1b[16], w2a[16], w2b[16], wcr[8];
  auto wfrag = [&](const bf16_t* wi, const bf16_t* wh, int c16_, int q_, int j, int s) {
    return *reinterpret_cast<const bf16x8*>((wave < 2 ? wi : wh) + (size_t)((c16_ & 3) * HD + 16 * member + 4 * j + (c16_ >> 2)) * HD + 256 * (wave & 1) + 32 * s + 8 * q_);
  };
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      w1a[j * 8 + s] = wfrag(p.w1i, p.w1h, c16, q, j, s); w1b[j * 8 + s] = wfrag(p.w1i, p.w1h, c16, q, j + 2, s);
      w2a[j * 8 + s] = wfrag(p.w2i, p.w2h, c16, q, j, s); w2b[j * 8 + s] = wfrag(p.w2i, p.w2h, c16, q, j + 2, s);
    }
#pragma unroll
  for (int s = 0; s < 8; ++s) wcr[s] = *reinterpret_cast<const bf16x8*>(p.wc + (size_t)(16 * member + c16) * 2 * HD + 256 * wave + 32 * s + 8 * q);
  float b2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b2[i] = p.b2i[i * HD + unit] + p.b2h[i * HD + unit];
  // zx1 (gate input of layer 1: embedding part + biases) of the NEXT step and, with the per-token table, the token after it: fetched by LDS-DMA in P3
  // (a compiler-tracked load would put an s_waitcnt vmcnt(0) -- which also waits for the phase's young stores -- in front of its first use)
  float c1[NCH], c2[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int row = min(row0 + RC * c + c16, B - 1);
    c1[c] = p.cs[0][(size_t)row * HD + unit]; c2[c] = p.cs[1][(size_t)row * HD + unit];
    const size_t zrow = p.zx_tok ? (size_t)(min(max(p.zx_tok[(int64_t)row * p.zx_sb], 1), p.V) - 1) : (size_t)row;
#pragma unroll
    for (int i = 0; i < 4; ++i) zxs[(c * 5 + i) * 256 + tid] = p.zx1[zrow * 4 * HD + i * HD + unit];
    reinterpret_cast<int*>(zxs)[(c * 5 + 4) * 256 + tid] = p.zx_tok ? p.zx_tok[(int64_t)min(1, L - 1) * p.zx_st + (int64_t)row * p.zx_sb] : 1;
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    load_rows16(p.out_b, row0 + RC * c, B, lds + (size_t)(c * 3 + 0) * OPB, tid);
    load_rows16(p.hsb[0], row0 + RC * c, B, lds + (size_t)(c * 3 + 1) * OPB, tid);
    load_rows16(p.hsb[1], row0 + RC * c, B, lds + (size_t)(c * 3 + 2) * OPB, tid);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): nothing of the prologue is in flight inside the loop
  __syncthreads();
  [[maybe_unused]] u64 stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();

  const bf16_t* const ca = p.ctxa + (size_t)min(arow, B - 1) * T * HD;
  const bf16_t* const cx = p.ctxb + (size_t)min(arow, B - 1) * T * HD;
  const int ntile = (T + 15) >> 4;
  bf16x8 cavr[RES ? 16 : 1];
  if constexpr (RES) {
    const bf16_t* carow = ca + (size_t)min(16 * wave + c16, T - 1) * HD + 8 * q;
#pragma unroll
    for (int s = 0; s < 16; ++s) cavr[s] = *reinterpret_cast<const bf16x8*>(carow + 32 * s);
  }

  // one K-split product of a chain: acc tiles -> LDS -> this wave's tile summed over the four waves (dec_cluster.hip's order)
  auto product = [&](const bf16x8 (&wa)[16], const bf16x8 (&wb)[16], const unsigned char* x0, const unsigned char* x1, f32x4& v, auto&& mid) {
    const unsigned char* src = (wave < 2 ? x0 : x1) + (256 * (wave & 1) + 8 * q) * 2 + (size_t)c16 * PA;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(src + 64 * s);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? wa[j * 8 + s] : wb[(j - 2) * 8 + s], bv, acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(red + ((size_t)(wave * 4 + j) * 64 + lane) * 4) = acc[j];
    mid();                                                          // the next phase's operand fetch: its producers published a phase ago, ~half a phase before its use
    lds_barrier();
    v = *reinterpret_cast<const f32x4*>(red + ((size_t)(0 * 4 + wave) * 64 + lane) * 4);
#pragma unroll
    for (int w2_ = 1; w2_ < 4; ++w2_) v += *reinterpret_cast<const f32x4*>(red + ((size_t)(w2_ * 4 + wave) * 64 + lane) * 4);
  };
  // LSTM cell on this lane's (unit, row): returns the packed h of the four units of (row, wave) in lanes q == 0
  auto cell = [&](const f32x4& z, float& c, f32x4& g, u32x2& hp) {
    const float ig = sigmoidf_(z[0]), fg = sigmoidf_(z[1]), og = sigmoidf_(z[2]), gg = tanhf_(z[3]);
    const float cn = fg * c + ig * gg, hn = og * tanhf_(cn);
    c = cn; g = f32x4{ig, fg, og, gg};
    const unsigned hb = bfbits(hn);
    const unsigned h1v = __shfl(hb, lane + 16, 64), h2v = __shfl(hb, lane + 32, 64), h3v = __shfl(hb, lane + 48, 64);
    hp = u32x2{sane(hb | (h1v << 16)), sane(h2v | (h3v << 16))};
  };
  bool dead = false;

  for (int t = 0; t < L && !dead; ++t) {
    int ot = tid; asm volatile("" : "+v"(ot));                     // opaque per-step copy of the thread id: the address arithmetic stays inside the step
    const int oc16 = ot & 15, oq = (ot >> 4) & 3, olane = ot & 63, ounit = 16 * member + 4 * wave + oq;
    unsigned char* const otrash = trash0 + ot * 16;

    // =================== P1: layer 1.  stores after the prefetch issue: publish + gates + cell state = 3
    auto P1 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const F = lds + (size_t)(c * 3 + 0) * OPB; unsigned char* const H1 = lds + (size_t)(c * 3 + 1) * OPB;
      const int rb = row0 + RC * c;
      if (t > 0) {
        if constexpr (c == 0) pend_land<3>(p.out_b + (size_t)t * slot, HD * 2, rb, B, F, PA, olane, wave, member, local, p.err, 11, &s_dead);      // P4<1> of step t-1: 2 stores + the pre-fill
        else pend_land<3>(p.out_b + (size_t)t * slot, HD * 2, rb, B, F, PA, olane, wave, member, local, p.err, 11, &s_dead);                       // P1<0>: 3 stores
      } else lds_barrier();
      f32x4 z, g; u32x2 hp;
      product(w1a, w1b, F, H1, z, [&] {
        if constexpr (c == 0) { if (t > 0) pend_issue(p.out_b + (size_t)t * slot, HD * 2, row0 + RC, B, lds + (size_t)(1 * 3 + 0) * OPB, PA, olane, wave, member, local); }
        else pend_issue(p.hsb[0] + (size_t)(t + 1) * slot, HD * 2, row0, B, lds + (size_t)(0 * 3 + 1) * OPB, PA, olane, wave, member, local);
      });
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] += zxs[(c * 5 + i) * 256 + ot];
      cell(z, c1[c], g, hp);
      const int row = rb + oc16; const bool ok = row < B;
      pst8(oq == 0 && ok ? (void*)(p.hsb[0] + (size_t)(t + 1) * slot + (size_t)row * HD + 16 * member + 4 * wave) : (void*)otrash, hp, local);
      st16f(ok && p.gates[0] ? (void*)(p.gates[0] + (((size_t)t * B + row) * HD + ounit) * 4) : (void*)otrash, g);
      st4f(ok ? (void*)(p.cs[0] + (size_t)(t + 1) * slot + (size_t)row * HD + ounit) : (void*)otrash, c1[c]);
    };
    // =================== P2: layer 2.  stores: publish + gates + cell state + the h half of [c ; h2] = 4
    auto P2 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const H1 = lds + (size_t)(c * 3 + 1) * OPB; unsigned char* const H2 = lds + (size_t)(c * 3 + 2) * OPB;
      const int rb = row0 + RC * c;
      if constexpr (c == 0) pend_land<3>(p.hsb[0] + (size_t)(t + 1) * slot, HD * 2, rb, B, H1, PA, olane, wave, member, local, p.err, 12, &s_dead);     // behind P1<1>
      else pend_land<4>(p.hsb[0] + (size_t)(t + 1) * slot, HD * 2, rb, B, H1, PA, olane, wave, member, local, p.err, 12, &s_dead);                       // behind P2<0>
      f32x4 z, g; u32x2 hp;
      product(w2a, w2b, H1, H2, z, [&] {
        if constexpr (c == 0) pend_issue(p.hsb[0] + (size_t)(t + 1) * slot, HD * 2, row0 + RC, B, lds + (size_t)(1 * 3 + 1) * OPB, PA, olane, wave, member, local);
        else pend_issue(p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, row0, B, lds + (size_t)(0 * 3 + 2) * OPB, PA, olane, wave, member, local);
      });
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] += b2[i];
      cell(z, c2[c], g, hp);
      const int row = rb + oc16; const bool ok = row < B;
      pst8(oq == 0 && ok ? (void*)(p.hsb[1] + (size_t)(t + 1) * slot + (size_t)row * HD + 16 * member + 4 * wave) : (void*)otrash, hp, local);
      st16f(ok && p.gates[1] ? (void*)(p.gates[1] + (((size_t)t * B + row) * HD + ounit) * 4) : (void*)otrash, g);
      st4f(ok ? (void*)(p.cs[1] + (size_t)(t + 1) * slot + (size_t)row * HD + ounit) : (void*)otrash, c2[c]);
      st8(oq == 0 && ok ? (void*)(p.cat_b + ((size_t)t * B + row) * 2 * HD + HD + 16 * member + 4 * wave) : (void*)otrash, hp);               // JoinTable [c ; h_top], LSTM.lua:153
    };
    // =================== P3: attention of row `member` (its chain's phase only).  stores: owners publish c + a = 2, the others none
    auto P3 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const H2 = lds + (size_t)(c * 3 + 2) * OPB;
      const int rb = row0 + RC * c;
      if constexpr (c == 0) pend_land<4>(p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, rb, B, H2, PA, olane, wave, member, local, p.err, 13, &s_dead);      // behind P2<1>
      else { if (mych == 0) pend_land<2>(p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, rb, B, H2, PA, olane, wave, member, local, p.err, 13, &s_dead);      // behind P3<0>
             else pend_land<0>(p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, rb, B, H2, PA, olane, wave, member, local, p.err, 13, &s_dead); }
      {                                                             // zx1 of the next step (LDS-DMA, older than the prefetch below: complete by the next counted wait)
        const int tn = min(t + 1, L - 1);
        const int row = min(rb + oc16, B - 1);
        const int ntok = reinterpret_cast<const int*>(zxs)[(c * 5 + 4) * 256 + ot];
        const size_t zr = p.zx_tok ? (size_t)(min(max(ntok, 1), p.V) - 1) : (size_t)tn * B + row;
        const unsigned zb = __builtin_amdgcn_readfirstlane(lds_addr(zxs) + (c * 5 * 256 + wave * 64) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma4(p.zx1 + zr * 4 * HD + i * HD + ounit, zb + i * 1024);
        if (p.zx_tok) dma4(p.zx_tok + (int64_t)min(t + 2, L - 1) * p.zx_st + (int64_t)row * p.zx_sb, zb + 4 * 1024);
      }
      auto fetch_next = [&] {
        if constexpr (c == 0) pend_issue(p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, row0 + RC, B, lds + (size_t)(1 * 3 + 2) * OPB, PA, olane, wave, member, local);
        else pend_issue(p.cat_b + (size_t)t * B * 2 * HD, HD * 4, row0, B, lds + (size_t)(0 * 3 + 0) * OPB, PA, olane, wave, member, local);
      };
      if (mych != c) { fetch_next(); return; }
      const unsigned char* hrow = H2 + (size_t)(member & (RC - 1)) * PA + 16 * q;
      if constexpr (RES) {
        if (wave < ntile) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cavr[s], *reinterpret_cast<const bf16x8*>(hrow + 64 * s), acc, 0, 0, 0);
          if (c16 == 0) *reinterpret_cast<f32x4*>(sc + 16 * wave + 4 * q) = acc;
        }
      } else {
        bf16x8 cav[16];                                             // tile `wave` of the pre-multiplied context of this member's row
        {
          const bf16_t* carow = ca + (size_t)min(16 * wave + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
          for (int s = 0; s < 16; ++s) cav[s] = *reinterpret_cast<const bf16x8*>(carow + 32 * s);
        }
        for (int tile = wave; tile < ntile; tile += 4) {
          if (tile != wave) {
            const bf16_t* r2 = ca + (size_t)min(16 * tile + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
            for (int s = 0; s < 16; ++s) cav[s] = *reinterpret_cast<const bf16x8*>(r2 + 32 * s);
          }
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cav[s], *reinterpret_cast<const bf16x8*>(hrow + 64 * s), acc, 0, 0, 0);
          if (c16 == 0) *reinterpret_cast<f32x4*>(sc + 16 * tile + 4 * q) = acc;
        }
      }
      bf16x8 cv[16];                                                // the first 64 context rows of the weighted sum: in flight across the softmax
#pragma unroll
      for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + (size_t)min(4 * i + wave, T - 1) * HD + 8 * olane);
      fetch_next();
      lds_barrier();
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
      const float av = wave == 0 ? aj[0] : wave == 1 ? aj[1] : wave == 2 ? aj[2] : aj[3];      // a[tid]
      float cacc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) cacc[e] = 0.f;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (64 * ch >= T) break;
        if (ch > 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + (size_t)min(64 * ch + 4 * i + wave, T - 1) * HD + 8 * olane);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, aj[ch]), 4 * i + wave));     // 0 beyond T
#pragma unroll
          for (int e = 0; e < 8; ++e) cacc[e] = fmaf(a, (float)cv[i][e], cacc[e]);
        }
      }
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8) = f32x4{cacc[0], cacc[1], cacc[2], cacc[3]};
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8 + 4) = f32x4{cacc[4], cacc[5], cacc[6], cacc[7]};
      lds_barrier();
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { v0 += part[w * HD + 2 * tid]; v1 += part[w * HD + 2 * tid + 1]; }
      pst4(rvalid ? (void*)(p.cat_b + ((size_t)t * B + arow) * 2 * HD + 2 * ot) : (void*)otrash, sane(bfpair(v0, v1)), local);      // c of row `member`: units 2 tid, 2 tid + 1
      st4f(rvalid && ot < T ? (void*)(p.a_all + ((size_t)t * B + arow) * T + ot) : (void*)otrash, av);
    };
    // =================== P4: out = tanh(W_c [c ; h2]), LSTM.lua:153-157.  stores: publish + the fp32 copy = 2 (+ the pre-fill behind chain 1)
    auto P4 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const F = lds + (size_t)(c * 3 + 0) * OPB; unsigned char* const H2 = lds + (size_t)(c * 3 + 2) * OPB;
      const int rb = row0 + RC * c;
      if constexpr (c == 0) { if (mych == 1) pend_land<2>(p.cat_b + (size_t)t * B * 2 * HD, HD * 4, rb, B, F, PA, olane, wave, member, local, p.err, 14, &s_dead);      // behind P3<1>
                              else pend_land<0>(p.cat_b + (size_t)t * B * 2 * HD, HD * 4, rb, B, F, PA, olane, wave, member, local, p.err, 14, &s_dead); }
      else pend_land<2>(p.cat_b + (size_t)t * B * 2 * HD, HD * 4, rb, B, F, PA, olane, wave, member, local, p.err, 14, &s_dead);                                          // behind P4<0>
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* src = (wave < 2 ? F : H2) + (256 * (wave & 1) + 8 * q) * 2 + (size_t)c16 * PA;   // k = 256 wave + 32 s: waves 0, 1 read c, waves 2, 3 read h2
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcr[s], *reinterpret_cast<const bf16x8*>(src + 64 * s), acc, 0, 0, 0);
      *reinterpret_cast<f32x4*>(red + ((size_t)wave * 64 + lane) * 4) = acc;
      if constexpr (c == 0) pend_issue(p.cat_b + (size_t)t * B * 2 * HD, HD * 4, row0 + RC, B, lds + (size_t)(1 * 3 + 0) * OPB, PA, olane, wave, member, local);
      else { if (t + 1 < L) pend_issue(p.out_b + (size_t)(t + 1) * slot, HD * 2, row0, B, lds + (size_t)(0 * 3 + 0) * OPB, PA, olane, wave, member, local); }
      lds_barrier();
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave == 0) {
        v = *reinterpret_cast<const f32x4*>(red + ((size_t)0 * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((size_t)w * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tanhf_(v[i]);
      }
      const int row = rb + oc16; const bool ok = wave == 0 && row < B;
      const size_t o = (size_t)(t + 1) * slot + (size_t)row * HD + 16 * member + 4 * oq;
      pst8(ok ? (void*)(p.out_b + o) : (void*)otrash, u32x2{sane(bfpair(v[0], v[1])), sane(bfpair(v[2], v[3]))}, local);
      st16f(ok ? (void*)(p.out + o) : (void*)otrash, v);
    };

    P1(IC<0>{}); CH_STAMP(0); P1(IC<1>{}); CH_STAMP(1); if (s_dead) { dead = true; break; }
    P2(IC<0>{}); CH_STAMP(2); P2(IC<1>{}); CH_STAMP(3); if (s_dead) { dead = true; break; }
    P3(IC<0>{}); CH_STAMP(4); P3(IC<1>{}); CH_STAMP(5); if (s_dead) { dead = true; break; }
    P4(IC<0>{}); CH_STAMP(6); P4(IC<1>{}); CH_STAMP(7); if (s_dead) { dead = true; break; }
    prefill(t + 2, ot, local);
  }
  wait_vm<0>();
#ifdef DC_DEBUG_STAMPS
  if (p.stamps && wid == 0 && tid == 0) { stamp[15] = local ? 1 : 0; for (int k = 0; k < 16; ++k) p.stamps[k] = stamp[k]; }
#endif
}

// ---------------------------------------------------------------------------------------------
bool dec_chain_enabled() { const char* e = getenv("AOCR_NO_DEC_CHAINS"); return !(e && e[0] == '1'); }

void dec_chain_forward(hipStream_t s, const DecClFwdArgs& a0) {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_FWD_LDS);
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_FWD_LDS);
  const bool res = a0.T <= 64 && !getenv("AOCR_CH_NO_RES");
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClFwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    a.stamps = getenv("AOCR_DC_STAMPS") ? a.xtab + (size_t)groups * NM : nullptr;
    if (res) hipLaunchKernelGGL((dec_ch_fwd_kernel<false, true>), dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), (size_t)CH_FWD_LDS, s, a);
    else hipLaunchKernelGGL((dec_ch_fwd_kernel<false, false>), dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), (size_t)CH_FWD_LDS, s, a);
  }
}

}  // namespace aocr
