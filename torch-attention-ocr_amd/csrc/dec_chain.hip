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
constexpr int CH_FWD_LDS = NCH * 3 * OPB + 16384 + 8192 + 1024 + NCH * 5 * 1024;
constexpr int CH_BEAM_EXTRA = 2048;                                 // beam search: candidate scores of the member's image + per-beam score / token / trie node
constexpr int CH_DEC_LDS = CH_FWD_LDS + 2560 + NCH * 1024 + NCH * 1024 + 10240;      // greedy decode: + W_o slice, out(t) of the member's units, token staging, the per-token gate-input table slice     // operands + K-split partial tiles + attention partial context + scores + the next step's gate inputs / tokens

__device__ __forceinline__ unsigned sane(unsigned x) { return x == SENT ? 0xFFFEFFFFu : x; }
// unsigned maximum spelled out: the pattern test must never become a signed comparison (0xFFFFFFFF is -1), whatever overload set `max` finds
__device__ __forceinline__ unsigned umax(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned umax4(const u32x4& v) { return umax(umax(v[0], v[1]), umax(v[2], v[3])); }
__device__ __forceinline__ void pst16(void* p, u32x4 v, bool local) {
  if (local) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

#ifdef DC_DEBUG_STAMPS
#define CH_STAMP(k) do { if (p.stamps) { const u64 now_ = __builtin_readcyclecounter(); stamp[k] += now_ - tprev; tprev = now_; } } while (0)
#define CH_STAMP2(k) do { if (p.stamps) { const u64 now_ = __builtin_readcyclecounter(); stamp2[k] += now_ - tprev2; tprev2 = now_; } } while (0)
#else
#define CH_STAMP(k) do { } while (0)
#define CH_STAMP2(k) do { } while (0)
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
                                          int* err, int code, int* dead_flag, int chunk0 = 0, [[maybe_unused]] int* retries = nullptr) {
  wait_vm<NST>();
  u32x4 g[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) g[j] = *reinterpret_cast<const u32x4*>(dst + (size_t)((wave + 4 * j + member) & (RC - 1)) * pitch + chunk0 + lane * 16);
  unsigned mx = umax(umax(umax4(g[0]), umax4(g[1])), umax(umax4(g[2]), umax4(g[3])));
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
      mx = umax(umax(umax4(g[0]), umax4(g[1])), umax(umax4(g[2]), umax4(g[3])));
      if (!__any(mx == SENT)) break;
    }
#ifdef DC_DEBUG_STAMPS
    if (retries) *retries += spins;
#endif
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
// Eight B fragments (16 bytes each, 64 bytes apart) in ONE asm statement: the reads are issued back to back and waited for once -- hipcc
// otherwise emits read / wait / 4 MFMAs per k-step (~170 cycles each instead of 64).  One statement, so every output is valid when it ends.
struct Frag8 { bf16x8 v[8]; };
__device__ __forceinline__ void lds_read8(Frag8& f, const unsigned char* p) {
  const unsigned a = lds_addr(p);
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\tds_read_b128 %3, %8 offset:192\n\t"
               "ds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %8 offset:320\n\tds_read_b128 %6, %8 offset:384\n\tds_read_b128 %7, %8 offset:448\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(f.v[0]), "=&v"(f.v[1]), "=&v"(f.v[2]), "=&v"(f.v[3]), "=&v"(f.v[4]), "=&v"(f.v[5]), "=&v"(f.v[6]), "=&v"(f.v[7]) : "v"(a) : "memory");
}
template <int C> struct IC { static constexpr int value = C; };
}  // namespace

// =============================================================================================================================
// Forward: teacher-forced loop (train step, gold pass).  Phases of step t, each for chain 0 then chain 1 (operand <- published in):
//   P1  z1 = [out(t-1) ; h1(t-1)] W1^T + zx1(t) -> c1, h1(t)          out(t-1) <- P4 of step t-1        publishes h1(t)
//   P2  z2 = [h1(t) ; h2(t-1)] W2^T + b         -> c2, h2(t)          h1(t)    <- P1                    publishes h2(t)
//   P3  attention of row r on member r (members 16 c .. 16 c + 15 work for chain c, the others pass)     h2(t) <- P2      publishes c(t) of the row
//   P4  out(t) = tanh(W_c [c(t) ; h2(t)])                              c(t)     <- P3                    publishes out(t)
// BEAM (with DEC): beam search, model.lua:360-536.  A chain carries 16 / k images x k hypotheses (row = image * k + beam); the step slots of the
// exchange buffers are a rolling window of four (rows of this launch's groups only); the state gather by parent beam (model.lua:516-535) is
// a lane permutation of the OLD-state partial products and of the cell states (nothing moves in memory); the image's owner (member of its first
// row) runs project_select_kernel's selection over the k rows and publishes token + parent of every new row; history for beam_backtrace.
template <bool DEC, bool RES, bool BEAM = false>     // RES: T <= 64 -- the 16-step tile of ctx . W_a a wave multiplies stays in its registers for the whole loop (64 VGPRs)
__global__ __launch_bounds__(256, 1) void dec_ch_fwd_kernel(DecClFwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  [[maybe_unused]] const u64 t_kernel0 = __builtin_readcyclecounter();
  [[maybe_unused]] u64 t_real0 = 0;
#ifdef DC_DEBUG_STAMPS
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_real0));
#endif
  float* const red = reinterpret_cast<float*>(lds + NCH * 3 * OPB);            // [4 waves][4 tiles][64 lanes][4]: K-split partial tiles of one chain
  float* const part = reinterpret_cast<float*>(lds + NCH * 3 * OPB + 16384);   // attention: [4 waves][512] partial context
  float* const sc = part + 4 * HD;                                             //            [256] scores
  float* const zxs = sc + 256;                                                 // [chain][4 gates + token][256 threads]: zx1 of the next step, the token after it (LDS-DMA)
  float* const wos = zxs + NCH * 5 * 256;                                      // DEC: [40][16] this member's slice of W_o (fp32)
  float* const outs = wos + 640;                                               //      [chain][16 rows][16] out(t) of this member's units (fp32)
  unsigned* const tokst = reinterpret_cast<unsigned*>(outs + NCH * 256);       //      [chain][4 waves][64] the tokens chosen a step ago, as fetched (tag << 8 | token)
  float* const ztab = reinterpret_cast<float*>(tokst + NCH * 256);             //      [40 tokens][4 gates][16] this member's columns of the per-token gate-input table
  float* const cand = ztab + 2560;                                             // BEAM: [8 beams][40] candidate scores of this member's image
  float* const bsc = cand + 320;                                               //       [2 parities][8] running scores, then last tokens, then trie nodes
  int* const bpt = reinterpret_cast<int*>(bsc + 16); int* const bnd = bpt + 16;
  __shared__ int s_local, s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % NM, gl = (i8 / NM) * 8 + xcd;
  if (gl >= p.ngroups) return;
  const int group = p.group0 + gl;
  const int B = p.B, T = p.T, L = p.L;
  const int kb = BEAM ? p.beam : 1, ipc = RC / kb;                  // BEAM: hypotheses per image, images per chain
  const int gx = BEAM ? gl : group;                                 // index of the group in the exchange buffers (BEAM: of this launch)
  const int row0 = gx * R;                                          // its first row there
  const int Rs = BEAM ? p.rows_slot : B;                            // rows of one step slot
  const int unit = 16 * member + 4 * wave + q;
  const size_t slot = (size_t)Rs * HD;
  // element offsets of step slots: h / out [slot s1 = step + 1] and [c ; h2] [step]; BEAM: four rolling slots (behind the B rows of the initial state)
  auto hof = [&](int s1) { return BEAM ? ((size_t)B + (size_t)(s1 & 3) * Rs) * HD : (size_t)s1 * slot; };
  auto cof = [&](int s_) { return (size_t)(BEAM ? (s_ & 3) : s_) * Rs * 2 * HD; };
  const int mych = member >> 4;                                     // the chain of the row whose attention this workgroup computes
  int nvc[NCH], rlim[NCH];                                          // valid rows of a chain; row limit of its operand fetches (rows beyond repeat the last valid one)
#pragma unroll
  for (int c = 0; c < NCH; ++c) nvc[c] = BEAM ? max(0, min(B - (group * NCH + c) * ipc, ipc)) * kb : max(0, min(B - (row0 + RC * c), RC));
#pragma unroll
  for (int c = 0; c < NCH; ++c) rlim[c] = BEAM ? (nvc[c] > 0 ? row0 + RC * c + nvc[c] : row0 + nvc[0]) : B;
  auto imgof = [&](int c, int lr) { return BEAM ? (group * NCH + c) * ipc + lr / kb : row0 + RC * c + lr; };      // image (context / label row) of local row lr of chain c
  const int arow = row0 + member; const bool rvalid = (member & (RC - 1)) < nvc[mych];     // the row whose attention this workgroup computes
  const int aimg = imgof(mych, member & (RC - 1));
  const bool bown = BEAM && rvalid && (member & (RC - 1)) % kb == 0;                       // BEAM: this member selects for its image
  unsigned char* const trash0 = reinterpret_cast<unsigned char*>(p.err + 16);

  // ---- pre-fill of this member's pieces of step s: wave 0 out(s) [slot s + 1], wave 1 / 2 h1(s) / h2(s) [slot s + 1], wave 3 the c half of
  // [c ; h2](s) of the member's row.  ONE store per wave (invalid -> trash slot).
  auto prefill = [&](int s, int ot, bool loc) {
    const int ln = ot & 63; unsigned char* const trash = trash0 + ot * 16;
    void* dst;
    if (wave < 3) {
      bf16_t* const base = wave == 0 ? p.out_b : p.hsb[wave - 1];
      const int row = row0 + (ln >> 1);
      dst = (s < L && (BEAM || row < B)) ? (void*)(base + hof(s + 1) + (size_t)row * HD + 16 * member + 8 * (ln & 1)) : (void*)trash;
    } else dst = (s < L && rvalid) ? (void*)(p.cat_b + cof(s) + (size_t)arow * 2 * HD + 8 * ln) : (void*)trash;
    unsigned sv = SENT; asm volatile("" : "+v"(sv));                 // (re-materialised per call: a loop-carried constant was spilled to scratch and re-loaded behind an s_waitcnt vmcnt(0))
    pst16(dst, u32x4{sv, sv, sv, sv}, loc);
  };
  prefill(0, tid, false); prefill(1, tid, false);
  constexpr int PSZ = 32 * 32 * 40;                                  // partial logits of a group and step parity: [row][source member][40]
  float* const pown = DEC ? p.pbuf + ((size_t)gx * 32 + member) * 32 * 40 : nullptr;      // ... this member's row (+ parity * ngroups_all * PSZ)
  const size_t ppar = (size_t)p.pgroups * PSZ;
  if constexpr (DEC) {                                               // this member's row of both parities: unwritten; its token slot: no token
    for (int i = tid; i < 2 * 320; i += 256) pst16(pown + (i / 320) * ppar + (i % 320) * 4, u32x4{SENT, SENT, SENT, SENT}, false);
    if (tid == 0) pst4(p.tokx + (size_t)gx * 32 + member, 0u, false);
  }
  wait_vm<0>();
  __syncthreads();
  // ---- co-location check (rnn_cluster.hip); it is also the point after which every member's pre-fill of steps 0 and 1 is in memory
  u64* const xt = p.xtab + (size_t)gx * NM;
  if (wave == 0) {                                                   // lane m polls member m's entry: one round trip instead of 32 in a row (~20 us of the prologue)
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u;
    if (lane == 0) stg64(xt + member, ((u64)p.epoch << 32) | (u64)(xcc + 1u));
    bool same = true;
    if (lane < NM) {
      u64 v; int spins = 0;
      while ((unsigned)((v = ldg64(xt + lane)) >> 32) != p.epoch) { if (++spins > DC_SPIN_LIMIT) { atomicExch(p.err, 15); same = false; break; } __builtin_amdgcn_s_sleep(2); }
      if ((unsigned)v != xcc + 1u) same = false;
    }
    const bool all_same = __ballot(same) == ~0ull;
    if (lane == 0) { s_local = all_same && !p.force_remote; s_dead = 0; }
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
  int tokc[NCH] = {1, 1};                                            // DEC: the token this lane's row (of each chain) feeds to the current step
  [[maybe_unused]] int plane[NCH] = {lane, lane};                    // BEAM: the lane that holds this lane's unit of the row's parent hypothesis
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int row = min(imgof(c, c16), B - 1);
    c1[c] = p.cs[0][(size_t)row * HD + unit]; c2[c] = p.cs[1][(size_t)row * HD + unit];
    if constexpr (DEC) tokc[c] = p.tok0[(size_t)row * p.tok0_stride];                  // the GO tokens
    else {
      const size_t zrow = p.zx_tok ? (size_t)(min(max(p.zx_tok[(int64_t)row * p.zx_sb], 1), p.V) - 1) : (size_t)row;
#pragma unroll
      for (int i = 0; i < 4; ++i) zxs[(c * 5 + i) * 256 + tid] = p.zx1[zrow * 4 * HD + i * HD + unit];
      reinterpret_cast<int*>(zxs)[(c * 5 + 4) * 256 + tid] = p.zx_tok ? p.zx_tok[(int64_t)min(1, L - 1) * p.zx_st + (int64_t)row * p.zx_sb] : 1;
    }
  }
  if constexpr (DEC) {
    for (int i = tid; i < 640; i += 256) { const int v = i >> 4, u = i & 15; wos[i] = v < p.V ? p.wo[(size_t)v * HD + 16 * member + u] : 0.f; }
    for (int i = tid; i < 2560; i += 256) {
      const int v = i >> 6, g = (i >> 4) & 3, u = i & 15;
      ztab[i] = v < p.V ? p.zx1[(size_t)v * 4 * HD + g * HD + 16 * member + u] : 0.f;
    }
  }
  [[maybe_unused]] float bo_lane = 0.f;                              // BEAM: the projector bias of class `lane`
  if constexpr (BEAM) bo_lane = lane < p.V ? p.bo[lane] : 0.f;
  int t_exit = -1; bool fin0 = false;                                // DEC: the step at which every row of the group had finished (early exit)
  float score = 0.f; int prev_tok = 0, node = 0;                     // DEC, wave 0 of the row's owner: running log-probability, last token, trie node (-use_dictionary)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if constexpr (BEAM) {                                            // every hypothesis of an image starts from the image's state (model.lua:388-398)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j, row = idx >> 6, ch = idx & 63; const size_t so = (size_t)min(imgof(c, row), B - 1) * HD + ch * 8;
        *reinterpret_cast<u32x4*>(lds + (size_t)(c * 3 + 0) * OPB + (size_t)row * PA + ch * 16) = *reinterpret_cast<const u32x4*>(p.out_b + so);
        *reinterpret_cast<u32x4*>(lds + (size_t)(c * 3 + 1) * OPB + (size_t)row * PA + ch * 16) = *reinterpret_cast<const u32x4*>(p.hsb[0] + so);
        *reinterpret_cast<u32x4*>(lds + (size_t)(c * 3 + 2) * OPB + (size_t)row * PA + ch * 16) = *reinterpret_cast<const u32x4*>(p.hsb[1] + so);
      }
    } else {
      load_rows16(p.out_b, row0 + RC * c, B, lds + (size_t)(c * 3 + 0) * OPB, tid);
      load_rows16(p.hsb[0], row0 + RC * c, B, lds + (size_t)(c * 3 + 1) * OPB, tid);
      load_rows16(p.hsb[1], row0 + RC * c, B, lds + (size_t)(c * 3 + 2) * OPB, tid);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): nothing of the prologue is in flight inside the loop
  __syncthreads();
  [[maybe_unused]] u64 stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
  [[maybe_unused]] const u64 t_loop0 = tprev;
  [[maybe_unused]] u64 stamp2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev2 = 0;
  [[maybe_unused]] u64 t_real1 = 0;
#ifdef DC_DEBUG_STAMPS
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_real1));
#endif

  const bf16_t* const ca = p.ctxa + (size_t)min(aimg, B - 1) * T * HD;      // (BEAM: the context is not replicated, model.lua:373)
  const bf16_t* const cx = p.ctxb + (size_t)min(aimg, B - 1) * T * HD;
  const int ntile = (T + 15) >> 4;
  bf16x8 cavr[RES ? 16 : 1];
  if constexpr (RES) {
    const bf16_t* carow = ca + (size_t)min(16 * wave + c16, T - 1) * HD + 8 * q;
#pragma unroll
    for (int s = 0; s < 16; ++s) cavr[s] = *reinterpret_cast<const bf16x8*>(carow + 32 * s);
  }

  // one K-split product of a chain: acc tiles -> LDS -> this wave's tile summed over the four waves (dec_cluster.hip's order)
  auto product_sum = [&](f32x4& v, int la, int lb) {                // la / lb: the lane whose column is read from the partial tiles of waves 0, 1 / 2, 3 (BEAM: the parent row)
    v = *reinterpret_cast<const f32x4*>(red + ((size_t)(0 * 4 + wave) * 64 + la) * 4);
    v += *reinterpret_cast<const f32x4*>(red + ((size_t)(1 * 4 + wave) * 64 + la) * 4);
    v += *reinterpret_cast<const f32x4*>(red + ((size_t)(2 * 4 + wave) * 64 + lb) * 4);
    v += *reinterpret_cast<const f32x4*>(red + ((size_t)(3 * 4 + wave) * 64 + lb) * 4);
  };
  auto product_mma = [&](const bf16x8 (&wa)[16], const bf16x8 (&wb)[16], const unsigned char* x0, const unsigned char* x1, auto&& mid) {
    const unsigned char* src = (wave < 2 ? x0 : x1) + (256 * (wave & 1) + 8 * q) * 2 + (size_t)c16 * PA;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag8 bf; lds_read8(bf, src);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? wa[j * 8 + s] : wb[(j - 2) * 8 + s], bf.v[s], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(red + ((size_t)(wave * 4 + j) * 64 + lane) * 4) = acc[j];
    mid();
    lds_barrier();
  };
  auto product = [&](const bf16x8 (&wa)[16], const bf16x8 (&wb)[16], const unsigned char* x0, const unsigned char* x1, f32x4& v, auto&& mid) {
    const unsigned char* src = (wave < 2 ? x0 : x1) + (256 * (wave & 1) + 8 * q) * 2 + (size_t)c16 * PA;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    Frag8 bf; lds_read8(bf, src);
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(j < 2 ? wa[j * 8 + s] : wb[(j - 2) * 8 + s], bf.v[s], acc[j], 0, 0, 0);
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
  [[maybe_unused]] int nretry[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // one-instruction stores of a phase behind its operand prefetch (what the next landing's counted wait leaves in flight).  Decode (DEC) keeps the state in
  // registers and reads nothing back: the saved gates, cell-state slots, attention weights and the fp32 out are not written at all -- a store instruction
  // costs ~250-400 cycles of issue on this path whatever it carries (tools/debug/enc_stamp.py), six per chain and step
  constexpr int S1 = DEC ? 1 : 3, S2 = DEC ? 2 : 4, S3 = DEC ? 1 : 2, S4 = DEC ? 4 : 2;

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
        if constexpr (c == 0) pend_land<S4 + 1>(p.out_b + hof(t), HD * 2, rb, rlim[c], F, PA, olane, wave, member, local, p.err, 11, &s_dead, 0, &nretry[0 + c]);      // P4<1> of step t-1: 2 stores (DEC: 5) + the pre-fill (P5's stores, wave 0 of an owner only, come on top)
        else pend_land<S1>(p.out_b + hof(t), HD * 2, rb, rlim[c], F, PA, olane, wave, member, local, p.err, 11, &s_dead, 0, &nretry[0 + c]);                       // P1<0>: 3 stores
      } else lds_barrier();
      CH_STAMP(8 + c);
      if constexpr (DEC) {                                           // the tokens chosen at step t-1 (published by the rows' owners in P5): fetched beside the products, read behind them
        if (t > 0) dma4(p.tokx + (size_t)gx * 32 + RC * c + (olane & 15), __builtin_amdgcn_readfirstlane(lds_addr(tokst) + ((c * 4 + wave) * 64) * 4));
      }
      f32x4 z, g; u32x2 hp;
      auto mid1 = [&] {
        if constexpr (c == 0) { if (t > 0) pend_issue(p.out_b + hof(t), HD * 2, row0 + RC, rlim[1], lds + (size_t)(1 * 3 + 0) * OPB, PA, olane, wave, member, local); }
        else pend_issue(p.hsb[0] + hof(t + 1), HD * 2, row0, rlim[0], lds + (size_t)(0 * 3 + 1) * OPB, PA, olane, wave, member, local);
      };
      if constexpr (BEAM) product_mma(w1a, w1b, F, H1, mid1); else product(w1a, w1b, F, H1, z, mid1);
      constexpr int TSH = BEAM ? 9 : 8;                              // a token word: tag << TSH | (BEAM: parent beam << 6) | token
      if constexpr (DEC) {
        if (t > 0) {
          wait_vm<4>();                                              // everything older than the four operand fetches above: the token fetch among it
          const unsigned want = (p.epoch * 4096u + (unsigned)t) & (0xFFFFFFFFu >> TSH);
          const bool need = !BEAM || (olane & 15) < nvc[c];          // (BEAM: nobody selects for the unused rows of a chain)
          unsigned tk = tokst[(c * 4 + wave) * 64 + olane];
          if (__any(need && (tk >> TSH) != want)) {
            int spins = 0;
#pragma nounroll
            while (true) {
              asm volatile("" : "+v"(spins));
              if (++spins > DC_SPIN_LIMIT) { if (olane == 0) { atomicExch(p.err, 17); s_dead = 1; } break; }
              __builtin_amdgcn_s_sleep(1);
              ld4_sc1(tk, (unsigned)((RC * c + (olane & 15)) * 4), p.tokx + (size_t)gx * 32, local);
              wait_vm<0>();
              asm volatile("" : "+v"(tk));
              if (!__any(need && (tk >> TSH) != want)) break;
            }
          }
          if constexpr (BEAM) {
            // model.lua:516-535: row r continues hypothesis `parent` of its image -- z1 and c1 (here), the h2(t-1) half of z2 and c2 (P2) are taken
            // from the parent's column / lane
            tokc[c] = need ? (int)(tk & 63u) : 1;
            const int pr = (oc16 / kb) * kb + (int)((tk >> 6) & 7u);
            plane[c] = need ? (olane & 48) | min(pr, RC - 1) : olane;
            c1[c] = __shfl(c1[c], plane[c], 64);
          } else tokc[c] = (int)(tk & 0xFFu);
        }
        if constexpr (BEAM) product_sum(z, plane[c], plane[c]);
        if constexpr (!BEAM) if (t > 0) {
          // Every row of the group has emitted EOS (or PAD): from here on each step selects PAD at no cost (model.lua:448-449), so the labels of
          // the remaining steps are PAD and the scores final.  Every wave of every member sees the same 32 tokens: all leave together (behind P1<1>).
          const bool done = tokc[c] == 1 || tokc[c] == 3 || rb + (olane & 15) >= B;
          const bool fin = __ballot(done) == ~0ull;
          if constexpr (c == 0) fin0 = fin; else { if (fin0 && fin && !p.no_early) t_exit = t; }
        }
        const int zrow = min(max(tokc[c], 1), p.V) - 1;             // (clamped: a corrupted token must not become a wild address)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] += ztab[zrow * 64 + i * 16 + 4 * wave + oq];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] += zxs[(c * 5 + i) * 256 + ot];
      }
      cell(z, c1[c], g, hp);
      const int row = rb + oc16; const bool ok = oc16 < nvc[c];
      pst8(oq == 0 && ok ? (void*)(p.hsb[0] + hof(t + 1) + (size_t)row * HD + 16 * member + 4 * wave) : (void*)otrash, hp, local);
      if constexpr (!DEC) {
        st16f(ok && p.gates[0] ? (void*)(p.gates[0] + (((size_t)t * B + row) * HD + ounit) * 4) : (void*)otrash, g);
        st4f(ok ? (void*)(p.cs[0] + (size_t)(t + 1) * slot + (size_t)row * HD + ounit) : (void*)otrash, c1[c]);
      }
    };
    // =================== P2: layer 2.  stores: publish + gates + cell state + the h half of [c ; h2] = 4
    auto P2 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const H1 = lds + (size_t)(c * 3 + 1) * OPB; unsigned char* const H2 = lds + (size_t)(c * 3 + 2) * OPB;
      const int rb = row0 + RC * c;
      if constexpr (c == 0) pend_land<S1>(p.hsb[0] + hof(t + 1), HD * 2, rb, rlim[c], H1, PA, olane, wave, member, local, p.err, 12, &s_dead, 0, &nretry[2 + c]);     // behind P1<1>
      else pend_land<S2>(p.hsb[0] + hof(t + 1), HD * 2, rb, rlim[c], H1, PA, olane, wave, member, local, p.err, 12, &s_dead, 0, &nretry[2 + c]);                       // behind P2<0>
      CH_STAMP(10 + c);
      f32x4 z, g; u32x2 hp;
      auto mid2 = [&] {
        if constexpr (c == 0) pend_issue(p.hsb[0] + hof(t + 1), HD * 2, row0 + RC, rlim[1], lds + (size_t)(1 * 3 + 1) * OPB, PA, olane, wave, member, local);
        else pend_issue(p.hsb[1] + hof(t + 1), HD * 2, row0, rlim[0], lds + (size_t)(0 * 3 + 2) * OPB, PA, olane, wave, member, local);
      };
      if constexpr (BEAM) {                                          // h1(t) belongs to the new rows, h2(t-1) and c2 to their parents
        product_mma(w2a, w2b, H1, H2, mid2);
        product_sum(z, olane, plane[c]);
        c2[c] = __shfl(c2[c], plane[c], 64);
      } else product(w2a, w2b, H1, H2, z, mid2);
#pragma unroll
      for (int i = 0; i < 4; ++i) z[i] += b2[i];
      cell(z, c2[c], g, hp);
      const int row = rb + oc16; const bool ok = oc16 < nvc[c];
      pst8(oq == 0 && ok ? (void*)(p.hsb[1] + hof(t + 1) + (size_t)row * HD + 16 * member + 4 * wave) : (void*)otrash, hp, local);
      if constexpr (!DEC) {
        st16f(ok && p.gates[1] ? (void*)(p.gates[1] + (((size_t)t * B + row) * HD + ounit) * 4) : (void*)otrash, g);
        st4f(ok ? (void*)(p.cs[1] + (size_t)(t + 1) * slot + (size_t)row * HD + ounit) : (void*)otrash, c2[c]);
      }
      st8(oq == 0 && ok ? (void*)(p.cat_b + cof(t) + (size_t)row * 2 * HD + HD + 16 * member + 4 * wave) : (void*)otrash, hp);               // JoinTable [c ; h_top], LSTM.lua:153
    };
    // =================== P3: attention of row `member` (its chain's phase only).  stores: owners publish c + a = 2, the others none
    auto P3 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const H2 = lds + (size_t)(c * 3 + 2) * OPB;
      const int rb = row0 + RC * c;
      if constexpr (c == 0) pend_land<S2>(p.hsb[1] + hof(t + 1), HD * 2, rb, rlim[c], H2, PA, olane, wave, member, local, p.err, 13, &s_dead, 0, &nretry[4 + c]);      // behind P2<1>
      else { if (mych == 0) pend_land<S3>(p.hsb[1] + hof(t + 1), HD * 2, rb, rlim[c], H2, PA, olane, wave, member, local, p.err, 13, &s_dead, 0, &nretry[4 + c]);      // behind P3<0>
             else pend_land<0>(p.hsb[1] + hof(t + 1), HD * 2, rb, rlim[c], H2, PA, olane, wave, member, local, p.err, 13, &s_dead, 0, &nretry[4 + c]); }
      CH_STAMP(12 + c);
      if constexpr (!DEC) {                                         // zx1 of the next step (LDS-DMA, older than the prefetch below: complete by the next counted wait)
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
        if constexpr (c == 0) pend_issue(p.hsb[1] + hof(t + 1), HD * 2, row0 + RC, rlim[1], lds + (size_t)(1 * 3 + 2) * OPB, PA, olane, wave, member, local);
        else pend_issue(p.cat_b + cof(t), HD * 4, row0, rlim[0], lds + (size_t)(0 * 3 + 0) * OPB, PA, olane, wave, member, local);
      };
      if (mych != c) { fetch_next(); return; }
      const unsigned char* hrow = H2 + (size_t)(member & (RC - 1)) * PA + 16 * q;
      if constexpr (RES) {
        if (wave < ntile) {
          f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            Frag8 bf; lds_read8(bf, hrow + 512 * hh);
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cavr[8 * hh + s], bf.v[s], acc, 0, 0, 0);
          }
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
          for (int hh = 0; hh < 2; ++hh) {
            Frag8 bf; lds_read8(bf, hrow + 512 * hh);
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cav[8 * hh + s], bf.v[s], acc, 0, 0, 0);
          }
          if (c16 == 0) *reinterpret_cast<f32x4*>(sc + 16 * tile + 4 * q) = acc;
        }
      }
      fetch_next();
      bf16x8 cv[16];                                                // the first 64 context rows of the weighted sum: in flight across the softmax
#pragma unroll
      for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + (size_t)min(4 * i + wave, T - 1) * HD + 8 * olane);
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
      pst4(rvalid ? (void*)(p.cat_b + cof(t) + (size_t)arow * 2 * HD + 2 * ot) : (void*)otrash, sane(bfpair(v0, v1)), local);      // c of row `member`: units 2 tid, 2 tid + 1
      if constexpr (!DEC) st4f(rvalid && ot < T ? (void*)(p.a_all + ((size_t)t * B + arow) * T + ot) : (void*)otrash, av);
    };
    // =================== P4: out = tanh(W_c [c ; h2]), LSTM.lua:153-157.  stores: publish + the fp32 copy = 2 (+ the pre-fill behind chain 1)
    auto P4 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const F = lds + (size_t)(c * 3 + 0) * OPB; unsigned char* const H2 = lds + (size_t)(c * 3 + 2) * OPB;
      const int rb = row0 + RC * c;
      if constexpr (c == 0) { if (mych == 1) pend_land<S3>(p.cat_b + cof(t), HD * 4, rb, rlim[c], F, PA, olane, wave, member, local, p.err, 14, &s_dead, 0, &nretry[6 + c]);      // behind P3<1>
                              else pend_land<0>(p.cat_b + cof(t), HD * 4, rb, rlim[c], F, PA, olane, wave, member, local, p.err, 14, &s_dead, 0, &nretry[6 + c]); }
      else pend_land<S4>(p.cat_b + cof(t), HD * 4, rb, rlim[c], F, PA, olane, wave, member, local, p.err, 14, &s_dead, 0, &nretry[6 + c]);                              // behind P4<0> (DEC: + 3 partial logits)
      CH_STAMP(14 + c);
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* src = (wave < 2 ? F : H2) + (256 * (wave & 1) + 8 * q) * 2 + (size_t)c16 * PA;   // k = 256 wave + 32 s: waves 0, 1 read c, waves 2, 3 read h2
      Frag8 bf; lds_read8(bf, src);
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wcr[s], bf.v[s], acc, 0, 0, 0);
      *reinterpret_cast<f32x4*>(red + ((size_t)wave * 64 + lane) * 4) = acc;
      if constexpr (c == 0) pend_issue(p.cat_b + cof(t), HD * 4, row0 + RC, rlim[1], lds + (size_t)(1 * 3 + 0) * OPB, PA, olane, wave, member, local);
      else { if (t + 1 < L) pend_issue(p.out_b + hof(t + 1), HD * 2, row0, rlim[0], lds + (size_t)(0 * 3 + 0) * OPB, PA, olane, wave, member, local); }
      lds_barrier();
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave == 0) {
        v = *reinterpret_cast<const f32x4*>(red + ((size_t)0 * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + ((size_t)w * 64 + lane) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = tanhf_(v[i]);
      }
      const int row = rb + oc16; const bool ok = wave == 0 && oc16 < nvc[c];
      const size_t o = hof(t + 1) + (size_t)row * HD + 16 * member + 4 * oq;
      pst8(ok ? (void*)(p.out_b + o) : (void*)otrash, u32x2{sane(bfpair(v[0], v[1])), sane(bfpair(v[2], v[3]))}, local);
      if constexpr (!DEC) st16f(ok ? (void*)(p.out + o) : (void*)otrash, v);
      if constexpr (DEC) {
        // projector (output_projector.lua:3-8) on fp32 out: this member's 16 units against its slice of W_o -> 16 rows x V partial logits, to the rows' owners
        if (wave == 0) *reinterpret_cast<f32x4*>(outs + c * 256 + c16 * 16 + 4 * q) = v;
        lds_barrier();
        const int r = ot >> 4, j = ot & 15;
        float o16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) o16[u] = outs[c * 256 + r * 16 + u];
        float* const pp = p.pbuf + (size_t)(t & 1) * ppar + ((size_t)gx * 32 + RC * c + r) * 32 * 40 + (size_t)member * 40;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int vv = j + 16 * k;
          float a = 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u) a = fmaf(wos[min(vv, 39) * 16 + u], o16[u], a);
          pst4(vv < 40 ? (void*)(pp + vv) : (void*)otrash, sane(__builtin_bit_cast(unsigned, a)), local);
        }
      }
    };
    // =================== P5 (greedy decode): LogSoftMax + selection of row `member` by wave 0 of its owner (project_select_kernel at beam 1; model.lua:376-536)
    auto P5 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (mych != c || wave != 0) return;
      const int V = p.V;
      float* const base = pown + (size_t)(t & 1) * ppar; const unsigned lo = (unsigned)(olane < V ? olane : 0) * 4u;
      float pv[32];
      int spins = 0;
#pragma nounroll
      while (true) {
#pragma unroll
        for (int sm = 0; sm < 32; ++sm) { unsigned u_; ld4_sc1(u_, (unsigned)(sm * 160) + lo, base, local); pv[sm] = __builtin_bit_cast(float, u_); }
        wait_vm<0>();
        unsigned mx = 0;
#pragma unroll
        for (int sm = 0; sm < 32; ++sm) { asm volatile("" : "+v"(pv[sm])); mx = umax(mx, __builtin_bit_cast(unsigned, pv[sm])); }
        if (!__any(mx == SENT)) break;
        asm volatile("" : "+v"(spins));
        if (++spins > DC_SPIN_LIMIT) { if (olane == 0) { atomicExch(p.err, 16); s_dead = 1; } break; }
        __builtin_amdgcn_s_sleep(1);
      }
      float x = -INFINITY;
      if (olane < V) {
        x = p.bo[olane];
#pragma unroll
        for (int sm = 0; sm < 32; ++sm) x += pv[sm];
      }
      const float mxv = wave_reduce(x, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
      const float sum = wave_reduce(olane < V ? expf(x - mxv) : 0.f, 0.f, [](float a, float b) { return a + b; });
      float lp = olane < V ? x - (mxv + logf(sum)) : -INFINITY;
      if (t > 0) {
        if (olane == 0 && (prev_tok == 1 || prev_tok == 3)) lp = 0.f;          // model.lua:448-449: after PAD / EOS only PAD, at no cost
        lp += score;                                                          // model.lua:450
      }
      unsigned long long tmask = 0;                                            // -use_dictionary: only tokens that continue the row's trie node (model.lua:413,469)
      if (p.trie_mask) {
        tmask = p.trie_mask[node];
        const bool okt = (t > 0 && olane == 0) || ((tmask >> olane) & 1ull);    // PAD is always admissible after the first step
        if (olane < V && !okt) lp = -INFINITY;
      }
      const float best = wave_reduce(lp, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
      const unsigned long long tie = __ballot(lp == best && olane < V);        // descending score, ties -> lowest index
      const int bi = tie ? __ffsll((long long)tie) - 1 : 0;
      score = best; prev_tok = bi + 1;
      if (p.trie_mask && !(t > 0 && bi == 0) && ((tmask >> bi) & 1ull))        // trie_next: PAD keeps the node (model.lua:502-503)
        node = p.trie_child[p.trie_base[node] + __popcll(tmask & ((1ull << bi) - 1ull))];
      if (olane == 0) {
        if (rvalid) { p.labels[(size_t)arow * p.tok0_stride + t] = bi + 1; if (t == L - 1) p.scores[arow] = best; }
        pst4(p.tokx + (size_t)gx * 32 + member, (((p.epoch * 4096u + (unsigned)(t + 1)) & 0xFFFFFFu) << 8) | (unsigned)(bi + 1), local);
      }
      // this parity's row is read: unwritten again for step t + 2 (in memory long before its writers get there: they need this step's token first)
      unsigned sv = SENT; asm volatile("" : "+v"(sv));
#pragma unroll
      for (int k = 0; k < 5; ++k) pst16(pown + (size_t)(t & 1) * ppar + (k * 64 + olane) * 4, u32x4{sv, sv, sv, sv}, local);
    };
    // =================== P5 (beam search): the owner of an image -- LogSoftMax of its k rows (wave j & 3 takes row j), then wave 0 selects the k best of the
    // k x V candidates exactly as project_select_kernel does (descending score, ties -> lowest index; model.lua:399-458) and publishes, for every new
    // row, token + parent beam; tokens and parents also go to the history beam_backtrace reads (model.lua:573-585)
    auto P5B = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (mych != c || !bown) return;
#ifdef DC_DEBUG_STAMPS
      tprev2 = __builtin_readcyclecounter();
#endif
      const int V = p.V, cur = t & 1, nxt = cur ^ 1;
      const bool first = t == 0;
      // partial logits of the image's k rows: [row][source member][40]; wave w sums source members 8 w .. 8 w + 7 of every row (lane = class; one batch of
      // up to 64 loads per wave), the four partial sums meet in LDS (`part`: idle outside P3)
      {
        const unsigned lo = (unsigned)(olane < V ? olane : 0) * 4u + (unsigned)wave * 8u * 160u;
        float* const base = pown + (size_t)(t & 1) * ppar;
        float pv[8][8];
        int spins = 0;
#pragma nounroll
        while (true) {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < kb) {
#pragma unroll
              for (int sm = 0; sm < 8; ++sm) { unsigned u_; ld4_sc1(u_, (unsigned)(j * 32 * 160 + sm * 160) + lo, base, local); pv[j][sm] = __builtin_bit_cast(float, u_); }
            }
          wait_vm<0>();
          unsigned mx = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j < kb) {
#pragma unroll
              for (int sm = 0; sm < 8; ++sm) { asm volatile("" : "+v"(pv[j][sm])); mx = umax(mx, __builtin_bit_cast(unsigned, pv[j][sm])); }
            }
          if (!__any(mx == SENT)) break;
          asm volatile("" : "+v"(spins));
          if (++spins > DC_SPIN_LIMIT) { if (olane == 0) { atomicExch(p.err, 16); s_dead = 1; } break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (j < kb) {
            float a = pv[j][0];
#pragma unroll
            for (int sm = 1; sm < 8; ++sm) a += pv[j][sm];
            part[(wave * 8 + j) * 64 + olane] = a;
          }
      }
      CH_STAMP2(0);
      lds_barrier();
      CH_STAMP2(1);
      for (int j = wave; j < kb; j += 4) {
        float x = -INFINITY;
        if (olane < V) x = bo_lane + (((part[(0 * 8 + j) * 64 + olane] + part[(1 * 8 + j) * 64 + olane]) + part[(2 * 8 + j) * 64 + olane]) + part[(3 * 8 + j) * 64 + olane]);
        const float mxv = wave_reduce(x, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
        const float sum = wave_reduce(olane < V ? expf(x - mxv) : 0.f, 0.f, [](float a, float b) { return a + b; });
        float lp = olane < V ? x - (mxv + logf(sum)) : -INFINITY;
        if (!first) {
          const int pt = bpt[cur * 8 + j];
          if (olane == 0 && (pt == 1 || pt == 3)) lp = 0.f;                     // model.lua:448-449: finished beams continue with PAD at zero cost
          lp += bsc[cur * 8 + j];                                               // model.lua:450
        }
        if (p.trie_mask) {                                                      // -use_dictionary (model.lua:413,469)
          const unsigned long long tm = p.trie_mask[first ? 0 : bnd[cur * 8 + j]];
          const bool okt = (!first && olane == 0) || ((tm >> olane) & 1ull);
          if (olane < V && !okt) lp = -INFINITY;
        }
        if (olane < V) cand[j * V + olane] = lp;
        unsigned sv = SENT; asm volatile("" : "+v"(sv));                        // this parity's row is read (by every wave: the barrier above): unwritten again for step t + 2
        float* const rowb = pown + (size_t)j * 32 * 40 + (size_t)(t & 1) * ppar;
#pragma unroll
        for (int k5 = 0; k5 < 5; ++k5) pst16(rowb + (k5 * 64 + olane) * 4, u32x4{sv, sv, sv, sv}, local);
      }
      CH_STAMP2(2);
      lds_barrier();
      CH_STAMP2(3);
      if (wave != 0) return;
      const int n = (first ? 1 : kb) * V;                                       // t = 0: one hypothesis per image (model.lua:388-398)
      // candidates c = lane + 64 i in registers; per pick: wave maximum (DPP), then the lowest index among the ties = lowest i first, lowest lane within it
      // (project_select_kernel's order: descending score, ties -> lowest index) -- no LDS traffic, no cross-lane permutes in the loop
      float cv[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) cv[i] = olane + 64 * i < n ? cand[olane + 64 * i] : -INFINITY;
      float first_best = -INFINITY; int first_bi = 0;
      float mysc = 0.f; int mybi = 0;
      for (int kk = 0; kk < kb; ++kk) {
        float lm = fmaxf(fmaxf(fmaxf(cv[0], cv[1]), fmaxf(cv[2], cv[3])), cv[4]);
        float best = wave_reduce(lm, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
        int bi = 0x7fffffff;
        if (best > -INFINITY) {                                                  // (a NaN or -inf row leaves bi unset: the fallback below)
#pragma unroll
          for (int i = 4; i >= 0; --i) { const unsigned long long tie = __ballot(cv[i] == best); if (tie) bi = 64 * i + __ffsll((long long)tie) - 1; }
        }
        if (bi == 0x7fffffff) { best = first_best; bi = first_bi; }             // model.lua:419-433 (see project_select_kernel)
        else {
#pragma unroll
          for (int i = 0; i < 5; ++i) if (bi == olane + 64 * i) cv[i] = -INFINITY;
        }
        if (kk == 0) { first_best = best; first_bi = bi; }
        if (olane == kk) { mysc = best; mybi = bi; }
      }
      CH_STAMP2(4);
      if (olane < kb) {
        const int tokv = mybi % V + 1, par = mybi / V;
        int node = 0;
        if (p.trie_mask) {                                                      // trie_next (model.lua:434-439,499-507)
          node = first ? 0 : bnd[cur * 8 + par];
          const int v0 = tokv - 1; const unsigned long long tm = p.trie_mask[node];
          if (!(!first && v0 == 0) && ((tm >> v0) & 1ull)) node = p.trie_child[p.trie_base[node] + __popcll(tm & ((1ull << v0) - 1ull))];
        }
        bsc[nxt * 8 + olane] = mysc; bpt[nxt * 8 + olane] = tokv; bnd[nxt * 8 + olane] = node;
        const size_t ho = ((size_t)t * B + aimg) * kb + olane;
        p.hist_tok[ho] = tokv; p.hist_par[ho] = par;
        if (t == L - 1) p.beam_scores[(size_t)aimg * kb + olane] = mysc;
        pst4(p.tokx + (size_t)gx * 32 + member + olane, (((p.epoch * 4096u + (unsigned)(t + 1)) & (0xFFFFFFFFu >> 9)) << 9) | ((unsigned)par << 6) | (unsigned)tokv, local);
      }
      CH_STAMP2(5);
    };

    P1(IC<0>{}); CH_STAMP(0); P1(IC<1>{}); CH_STAMP(1); if (s_dead) { dead = true; break; }
    if constexpr (DEC) { if (t_exit >= 0) break; }                   // (the PAD labels / final scores are written behind the loop)
    P2(IC<0>{}); CH_STAMP(2); P2(IC<1>{}); CH_STAMP(3); if (s_dead) { dead = true; break; }
    P3(IC<0>{}); CH_STAMP(4); P3(IC<1>{}); CH_STAMP(5); if (s_dead) { dead = true; break; }
    P4(IC<0>{}); CH_STAMP(6); P4(IC<1>{}); CH_STAMP(7); if (s_dead) { dead = true; break; }
    prefill(t + 2, ot, local);
    if constexpr (BEAM) { P5B(IC<0>{}); P5B(IC<1>{}); }
    else if constexpr (DEC) { P5(IC<0>{}); P5(IC<1>{}); }
  }
  if constexpr (DEC) {
    if (t_exit >= 0 && wave == 0 && lane == 0 && rvalid) {
      for (int tt = t_exit; tt < L; ++tt) p.labels[(size_t)arow * p.tok0_stride + tt] = 1;
      p.scores[arow] = score;
    }
  }
  wait_vm<0>();
#ifdef DC_DEBUG_STAMPS
  if (p.stamps && tid == 0) {      // per workgroup: start, loop start, end in 100 MHz ticks (low 31 bits)
    u64 rt; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt));
    p.err[16 + 2048 + 32 + wid] = (int)(t_real0 & 0x7FFFFFFF); p.err[16 + 2048 + 288 + wid] = (int)(t_real1 & 0x7FFFFFFF); p.err[16 + 2048 + 544 + wid] = (int)(rt & 0x7FFFFFFF);
  }
  if (p.stamps && wid == 0 && tid == 0) {
    for (int k = 0; k < 16; ++k) p.stamps[k] = stamp[k];
    for (int k = 0; k < 8; ++k) p.err[16 + 2048 + k] = nretry[k];
    for (int k = 0; k < 6; ++k) p.err[16 + 2048 + 12 + k] = (int)(stamp2[k] >> 4);
    p.err[16 + 2048 + 9] = (int)((t_loop0 - t_kernel0) >> 4); p.err[16 + 2048 + 10] = (int)((__builtin_readcyclecounter() - t_kernel0) >> 4);      // prologue, whole kernel (cycles / 16)
    u64 rt; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)); p.err[16 + 2048 + 11] = (int)(rt - t_real0);                       // whole kernel in 100 MHz ticks
  }
#endif
}

// =============================================================================================================================
// Backward (model.lua:643-661; dec_cluster.hip's dec_cl_bwd_kernel: same weight slices in registers, same products, same layouts).  Phases of
// step t (t = L-1 .. 0), each for chain 0 then chain 1; every operand was published one phase of the SAME chain earlier:
//   B2  [d c ; d h2a] = d pre W_c                               d pre <- B6 of step t+1 (the prologue for t = L-1)      publishes d c (fp32, to the rows' owners)
//   B3  attention backward of row r on member r (its chain's phase only)      the row's d c <- B2                          publishes d q (bf16 row)
//   B4  d h2 = d q W_a + d h2a + d h2rec; cell backward 2       d q <- B3                                                publishes d z2 (4 KB rows)
//   B5  [d h1 ; d h2rec] = d z2 [W2_i2h | W2_h2h]; cell backward 1            d z2 <- B4                                  publishes d z1
//   B6  [d h1rec ; d feed] = d z1 [W1_h2h | W1_i2h[:, E:]]; d pre(t-1) = (d out_proj + d feed)(1 - out^2)     d z1 <- B5       publishes d pre(t-1)
// The 4 KB d z rows are fetched by the wave that reads them (wave w: bytes 1024 w .. of every row = its quarter of K): no barrier between
// landing and use, and the look for unwritten dwords rides on the fragment reads.  What the elementwise parts read from global memory (saved gates
// and cell states, d out_proj, out) is fetched by LDS-DMA one phase ahead into small staging areas (wave 0 does the elementwise work of a chain).
constexpr int XC = RC * PZ;                                          // operand buffer of a chain: 16 rows x 4 KB (+ pad); d pre / d q rows use its start, pitch PA
constexpr int CH_BWD_LDS = NCH * XC + 8192 + 1024 + 2048 + NCH * 2048 + 2 * 6144 + NCH * 2048;
struct Frag4 { bf16x8 v[4]; };
__device__ __forceinline__ void lds_read4(Frag4& f, const unsigned char* p) {
  const unsigned a = lds_addr(p);
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:128\n\tds_read_b128 %3, %4 offset:192\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(f.v[0]), "=&v"(f.v[1]), "=&v"(f.v[2]), "=&v"(f.v[3]) : "v"(a) : "memory");
}
__device__ __forceinline__ unsigned fmax8(const Frag8& f) {
  unsigned m = 0;
#pragma unroll
  for (int s = 0; s < 8; ++s) m = umax(m, umax4(__builtin_bit_cast(u32x4, f.v[s])));
  return m;
}

__global__ __launch_bounds__(256, 1) void dec_ch_bwd_kernel(DecClBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  float* const red = reinterpret_cast<float*>(lds + NCH * XC);                 // [4 waves][2 tiles][64 lanes][4] partial tiles; attention: [4 waves][512] partial d q
  float* const part = red;
  float* const da = reinterpret_cast<float*>(lds + NCH * XC + 8192);           // [256] d a
  bf16_t* const dchl = reinterpret_cast<bf16_t*>(lds + NCH * XC + 8192 + 1024);        // [2][512]: the row's d c as hi + lo bf16
  unsigned char* const stin = lds + NCH * XC + 8192 + 1024 + 2048;                     // [chain][2][64 lanes][16 B]: d out_proj, out of the chain's next step (fetched in B5, read in B6)
  float* const dcs = reinterpret_cast<float*>(stin);                                    // [512]: the row's d c (fp32) as fetched (in B2 / B3, read in B3: stin is dead then)
  unsigned char* const cellb = stin + NCH * 2048;                                       // [2 buffers][6][64 lanes][16 B]: c(t), c(t-1), the four units' gates
  f32x4* const keep = reinterpret_cast<f32x4*>(cellb + 2 * 6144);                       // [chain][2][64 lanes]: wave 0's d h2a (B2 -> B4) and d h1rec (B6 -> B5 of the next step): LDS instead of 16 VGPRs
  __shared__ int s_local, s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % NM, gl = (i8 / NM) * 8 + xcd;
  if (gl >= p.ngroups) return;
  const int group = p.group0 + gl;
  const int B = p.B, T = p.T, L = p.L, row0 = group * R;
  const size_t slot = (size_t)B * HD;
  const int arow = row0 + member; const bool rvalid = arow < B;
  const int mych = member >> 4;
  unsigned char* const trash0 = reinterpret_cast<unsigned char*>(p.err + 16);

  // ---- pre-fill of this member's pieces of step s (3 stores per wave): d z2 / d z1 gate `wave`; wave 0 d pre, waves 1, 2 the halves of d c, wave 3 the row's d q
  auto prefill = [&](int s, int ot, bool loc) {
    const int ln = ot & 63; unsigned char* const trash = trash0 + ot * 16;
    unsigned sv = SENT; asm volatile("" : "+v"(sv));
    const u32x4 sent = u32x4{sv, sv, sv, sv};
    const bool on = s >= 0 && s < L;
    const int row = row0 + (ln >> 1); const bool rok = on && row < B;
#pragma unroll
    for (int l = 0; l < 2; ++l)
      pst16(rok ? (void*)(p.dzb[l] + ((size_t)s * B + row) * 4 * HD + wave * HD + 16 * member + 8 * (ln & 1)) : (void*)trash, sent, loc);
    void* dst;
    if (wave == 0) dst = rok ? (void*)(p.dpre_b + (size_t)s * slot + (size_t)row * HD + 16 * member + 8 * (ln & 1)) : (void*)trash;
    else if (wave == 3) dst = (on && rvalid) ? (void*)(p.dq_b + (size_t)s * slot + (size_t)arow * HD + 8 * ln) : (void*)trash;
    else { const int id = ln + 64 * (wave - 1), r2 = row0 + (id >> 2); dst = (on && r2 < B) ? (void*)(p.dcat + ((size_t)s * B + r2) * 2 * HD + 16 * member + 4 * (id & 3)) : (void*)trash; }
    pst16(dst, sent, loc);
  };
  prefill(L - 1, tid, false); prefill(L - 2, tid, false);
  wait_vm<0>();
  __syncthreads();
  u64* const xt = p.xtab + (size_t)group * NM;
  if (wave == 0) {                                                   // lane m polls member m's entry: one round trip instead of 32 in a row (~20 us of the prologue)
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u;
    if (lane == 0) stg64(xt + member, ((u64)p.epoch << 32) | (u64)(xcc + 1u));
    bool same = true;
    if (lane < NM) {
      u64 v; int spins = 0;
      while ((unsigned)((v = ldg64(xt + lane)) >> 32) != p.epoch) { if (++spins > DC_SPIN_LIMIT) { atomicExch(p.err, 25); same = false; break; } __builtin_amdgcn_s_sleep(2); }
      if ((unsigned)v != xcc + 1u) same = false;
    }
    const bool all_same = __ballot(same) == ~0ull;
    if (lane == 0) { s_local = all_same && !p.force_remote; s_dead = 0; }
  }
  __syncthreads();
  const bool local = __builtin_amdgcn_readfirstlane(s_local) != 0;

  // ---- resident weights: rows 16m + c16 of the transposed matrices (A fragments), this wave's quarter of K
  bf16x8 wz[4][16], wct[2][4], wat[4];
  {
    const bf16_t* wsrc[4] = {p.w2i_t, p.w2h_t, p.w1h_t, p.w1f_t};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int s = 0; s < 16; ++s) wz[k][s] = *reinterpret_cast<const bf16x8*>(wsrc[k] + (size_t)(16 * member + c16) * 4 * HD + 512 * wave + 32 * s + 8 * q);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      wct[0][s] = *reinterpret_cast<const bf16x8*>(p.wc_t + (size_t)(16 * member + c16) * HD + 128 * wave + 32 * s + 8 * q);
      wct[1][s] = *reinterpret_cast<const bf16x8*>(p.wc_t + (size_t)(HD + 16 * member + c16) * HD + 128 * wave + 32 * s + 8 * q);
      wat[s] = *reinterpret_cast<const bf16x8*>(p.wa_t + (size_t)(16 * member + c16) * HD + 128 * wave + 32 * s + 8 * q);
    }
  }
  const bf16_t* const cx = p.ctxb + (size_t)min(arow, B - 1) * T * HD;
  const int ntile = (T + 15) >> 4;
  // per chain, in wave 0 (lane -> row 16 c + c16 of the group, units 16 member + 4 q .. + 3): running cell-state gradients and what a step hands to the next one
  f32x4 dc1[NCH], dc2[NCH], dh2rec[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { dc1[c] = dc2[c] = dh2rec[c] = f32x4{0.f, 0.f, 0.f, 0.f}; if (wave == 0) keep[(c * 2 + 1) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  __builtin_amdgcn_s_waitcnt(0x0F70);

  // ---- operand fetches.  1 KB rows (d pre, d q): pend_issue / pend_land as in the forward kernel.  4 KB rows (d z): wave w fetches its own quarter of every row.
  auto xbuf = [&](int c) { return lds + (size_t)c * XC; };
  auto dz_issue = [&](const bf16_t* src, int rb, unsigned char* X, int ln) {
#pragma unroll
    for (int r = 0; r < RC; ++r)
      dma16x(reinterpret_cast<const unsigned char*>(src) + (size_t)min(rb + r, B - 1) * (HD * 8) + 1024 * wave + ln * 16, __builtin_amdgcn_readfirstlane(lds_addr(X) + r * PZ + 1024 * wave), local);
  };
  // wave 0's elementwise inputs of (chain c, step t): d out_proj(t), out(t)
  auto stin_issue = [&](int c, int t, int oc16, int oq, int ln) {
    if (wave != 0) return;
    const size_t eo = (size_t)min(row0 + RC * c + oc16, B - 1) * HD + 16 * member + 4 * oq;
    const unsigned b = __builtin_amdgcn_readfirstlane(lds_addr(stin) + c * 2048);
    (void)ln;
    dma16x(p.dout_proj + (size_t)t * slot + eo, b, true); dma16x(p.out + (size_t)(t + 1) * slot + eo, b + 1024, true);
  };
  // ... and the saved state of layer l: c(t), c(t-1), gates of the lane's four units
  auto cell_issue = [&](int buf, int c, int l, int t, int oc16, int oq) {
    if (wave != 0) return;
    const int row = min(row0 + RC * c + oc16, B - 1), u0 = 16 * member + 4 * oq;
    const unsigned b = __builtin_amdgcn_readfirstlane(lds_addr(cellb) + buf * 6144);
    dma16x(p.cs[l] + (size_t)(t + 1) * slot + (size_t)row * HD + u0, b, true);
    dma16x(p.cs[l] + (size_t)t * slot + (size_t)row * HD + u0, b + 1024, true);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16x(p.gates[l] + (((size_t)t * B + row) * HD + u0 + i) * 4, b + 2048 + 1024 * i, true);
  };
  auto reduce2 = [&](const f32x4 (&acc)[2], f32x4 (&v)[2], auto&& mid, auto&& post, bool release = false) {      // two K-split tiles of a chain -> wave 0 (dec_cluster.hip's order)
#pragma unroll
    for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x4*>(red + ((size_t)(wave * 2 + n) * 64 + lane) * 4) = acc[n];
    mid();
    lds_barrier();
    post();                                                          // fetches of 4 KB-row operands: as late as the phase allows for every wave (their producers published a phase ago; an early fetch finds unwritten dwords)
    if (wave == 0) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        v[n] = *reinterpret_cast<const f32x4*>(red + ((size_t)(0 * 2 + n) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v[n] += *reinterpret_cast<const f32x4*>(red + ((size_t)(w * 2 + n) * 64 + lane) * 4);
      }
    }
    // release: the NEXT phase has no landing barrier in front of its products (the d z phases), so its partial tiles could overwrite these before wave 0 has read them
    if (release) lds_barrier();
  };
  // cell backward of one layer on this lane's 4 units (EpGatesBwd), inputs from the staging buffer
  auto cell_bwd = [&](int buf, const f32x4& dh, f32x4& dcs_, f32x4 (&dz)[4]) {
    const unsigned char* b = cellb + buf * 6144 + lane * 16;
    const f32x4 cn = *reinterpret_cast<const f32x4*>(b), cp = *reinterpret_cast<const f32x4*>(b + 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(b + 2048 + 1024 * i);
      const float ig = g[0], fg = g[1], og = g[2], gg = g[3];
      const float tc = tanhf_(cn[i]);
      const float dcv = dh[i] * og * (1.f - tc * tc) + dcs_[i];
      const float d_o = dh[i] * tc, di = dcv * gg, dg = dcv * ig, df = dcv * cp[i];
      dz[0][i] = di * ig * (1.f - ig); dz[1][i] = df * fg * (1.f - fg); dz[2][i] = d_o * og * (1.f - og); dz[3][i] = dg * (1.f - gg * gg);
      dcs_[i] = dcv * fg;
    }
  };
  [[maybe_unused]] int zretry = 0;
  // K = 2048 products against a chain's d z (this wave's quarter of K, two tiles); the fragments are looked at for unwritten dwords on the way
  auto zprod = [&](const bf16x8 (&wa_)[16], const bf16x8 (&wb_)[16], const bf16_t* src, int rb, unsigned char* X, f32x4 (&v)[2], int code, auto&& mid, auto&& post) {
    const unsigned char* base = X + (size_t)c16 * PZ + 1024 * wave + 16 * q;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {                                 // (two halves: 32 instead of 64 fragment registers live beside the 304 of the weights)
      Frag8 f;
      lds_read8(f, base + 512 * hh);
      if (__any(fmax8(f) == SENT)) {
        int spins = 0;
#pragma nounroll
        while (true) {
          asm volatile("" : "+v"(spins));
          if (++spins > DC_SPIN_LIMIT) { if (lane == 0) { atomicExch(p.err, code); s_dead = 1; } break; }
          __builtin_amdgcn_s_sleep(1);
          {                                                          // lane (c16, q) holds row c16's fragments: re-fetch this wave's quarter of the rows that have unwritten dwords
            const unsigned long long bm = __ballot(fmax8(f) == SENT);
            unsigned rows = (unsigned)((bm | (bm >> 16) | (bm >> 32) | (bm >> 48)) & 0xFFFFull);
            while (rows) {
              const int r = __builtin_ctz(rows); rows &= rows - 1;
              dma16x(reinterpret_cast<const unsigned char*>(src) + (size_t)min(rb + r, B - 1) * (HD * 8) + 1024 * wave + lane * 16, __builtin_amdgcn_readfirstlane(lds_addr(X) + r * PZ + 1024 * wave), local);
            }
          }
          wait_vm<0>();
          lds_read8(f, base + 512 * hh);
          if (!__any(fmax8(f) == SENT)) break;
        }
#ifdef DC_DEBUG_STAMPS
        zretry += spins;
#endif
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa_[8 * hh + s], f.v[s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb_[8 * hh + s], f.v[s], acc[1], 0, 0, 0);
      }
    }
    reduce2(acc, v, mid, post, true);
  };
  // d pre of (chain c, step t) from d feed and the staged d out_proj / out: published (bf16) + stored (fp32) by wave 0 -- 2 stores per wave
  auto dpre_publish = [&](int c, int t, int ot, bool from_lds, const f32x4& dpn_, const f32x4& on_, const f32x4& dfeed) {
    const int oc16 = ot & 15, oq = (ot >> 4) & 3;
    f32x4 dpn = dpn_, on = on_;
    if (from_lds) { dpn = *reinterpret_cast<const f32x4*>(stin + c * 2048 + (ot & 63) * 16); on = *reinterpret_cast<const f32x4*>(stin + c * 2048 + 1024 + (ot & 63) * 16); }
    f32x4 dpre;
#pragma unroll
    for (int i = 0; i < 4; ++i) dpre[i] = (dpn[i] + dfeed[i]) * (1.f - on[i] * on[i]);
    const int erow = row0 + RC * c + oc16; const bool eok = wave == 0 && t >= 0 && erow < B;
    const size_t eoff = (size_t)min(erow, B - 1) * HD + 16 * member + 4 * oq;
    unsigned char* const otrash = trash0 + ot * 16;
    pst8(eok ? (void*)(p.dpre_b + (size_t)t * slot + eoff) : (void*)otrash, u32x2{sane(bfpair(dpre[0], dpre[1])), sane(bfpair(dpre[2], dpre[3]))}, local);
    st16f(eok ? (void*)(p.dpre + (size_t)t * slot + eoff) : (void*)otrash, dpre);
  };
  // ---- step L-1's d pre (d feed = 0) from direct loads
  {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const size_t eo = (size_t)min(row0 + RC * c + c16, B - 1) * HD + 16 * member + 4 * q;
      const f32x4 dpn = *reinterpret_cast<const f32x4*>(p.dout_proj + (size_t)(L - 1) * slot + eo), on = *reinterpret_cast<const f32x4*>(p.out + (size_t)L * slot + eo);
      dpre_publish(c, L - 1, tid, false, dpn, on, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    pend_issue(p.dpre_b + (size_t)(L - 1) * slot, HD * 2, row0, B, xbuf(0), PA, lane, wave, member, local);
  }
  [[maybe_unused]] u64 stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
  bool dead = false;
  bool first = true;

  for (int t = L - 1; t >= 0 && !dead; --t) {
    int ot = tid; asm volatile("" : "+v"(ot));
    const int olane = ot & 63, oc16 = ot & 15, oq = (ot >> 4) & 3;
    unsigned char* const otrash = trash0 + ot * 16;

    // =================== B2: [d c ; d h2a] = d pre W_c.  stores: the d c half (fp32) = 1
    auto B2 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const X = xbuf(c); const int rb = row0 + RC * c;
      if constexpr (c == 0) { if (first) pend_land<0>(p.dpre_b + (size_t)t * slot, HD * 2, rb, B, X, PA, olane, wave, member, local, p.err, 21, &s_dead);
                              else pend_land<5>(p.dpre_b + (size_t)t * slot, HD * 2, rb, B, X, PA, olane, wave, member, local, p.err, 21, &s_dead); }       // behind B6<1>: 2 stores + the pre-fill's 3
      else pend_land<1>(p.dpre_b + (size_t)t * slot, HD * 2, rb, B, X, PA, olane, wave, member, local, p.err, 21, &s_dead);                                    // behind B2<0>
      Frag4 f; lds_read4(f, X + (size_t)c16 * PA + (128 * wave + 8 * q) * 2);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dcat[2];
#pragma unroll
      for (int s = 0; s < 4; ++s) { acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wct[0][s], f.v[s], acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wct[1][s], f.v[s], acc[1], 0, 0, 0); }
      reduce2(acc, dcat, [&] {
        if constexpr (c == 0) pend_issue(p.dpre_b + (size_t)t * slot, HD * 2, row0 + RC, B, xbuf(1), PA, olane, wave, member, local);
        else { if (mych == 0 && wave < 2)                           // the d c row of a chain-0 owner (published in B2<0>)
                 dma16x(reinterpret_cast<const unsigned char*>(p.dcat + ((size_t)t * B + min(arow, B - 1)) * 2 * HD) + 1024 * wave + olane * 16, __builtin_amdgcn_readfirstlane(lds_addr(dcs) + 1024 * wave), local); }
      }, [] {});
      if (wave == 0) keep[(c * 2 + 0) * 64 + lane] = dcat[1];
      const int erow = rb + oc16; const bool eok = wave == 0 && erow < B;
      u32x4 dv = __builtin_bit_cast(u32x4, dcat[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) dv[i] = sane(dv[i]);
      pst16(eok ? (void*)(p.dcat + ((size_t)t * B + erow) * 2 * HD + 16 * member + 4 * oq) : (void*)otrash, dv, local);
    };
    // =================== B3: attention backward of row `member` (its chain's phase only).  stores: owners d q (bf16) + d s + d q (fp32) = 3
    auto B3 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      auto fetch_next = [&] {
        if constexpr (c == 0) { if (mych == 1 && wave < 2)           // the d c row of a chain-1 owner (published in B2<1>)
                                  dma16x(reinterpret_cast<const unsigned char*>(p.dcat + ((size_t)t * B + min(arow, B - 1)) * 2 * HD) + 1024 * wave + olane * 16, __builtin_amdgcn_readfirstlane(lds_addr(dcs) + 1024 * wave), local); }
        else { cell_issue(0, 0, 1, t, oc16, oq); pend_issue(p.dq_b + (size_t)t * slot, HD * 2, row0, B, xbuf(0), PA, olane, wave, member, local); }
      };
      if (mych != c) { fetch_next(); return; }
      // ---- the row's d c: 512 floats as staged by waves 0, 1 -> hi + lo bf16 in LDS
      if constexpr (c == 0) wait_vm<1>(); else wait_vm<0>();         // behind B2<1> (1 store) / an empty B3<0>
      lds_barrier();
      {
        f32x2 x = *reinterpret_cast<const f32x2*>(dcs + 2 * ot);
        if (__any(umax(__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[1])) == SENT)) {      // (this wave's half of the row: waves 0, 1 the first KB, 2, 3 the second)
          int spins = 0;
#pragma nounroll
          while (true) {
            asm volatile("" : "+v"(spins));
            if (++spins > DC_SPIN_LIMIT) { if (olane == 0) { atomicExch(p.err, 22); s_dead = 1; } break; }
            __builtin_amdgcn_s_sleep(1);
            u64 v; asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(v) : "v"((unsigned)(ot * 8)), "s"(p.dcat + ((size_t)t * B + min(arow, B - 1)) * 2 * HD) : "memory");
            wait_vm<0>();
            asm volatile("" : "+v"(v));
            x = f32x2{__builtin_bit_cast(float, (unsigned)v), __builtin_bit_cast(float, (unsigned)(v >> 32))};
            if (!__any(umax((unsigned)v, (unsigned)(v >> 32)) == SENT)) break;
          }
        }
        const float h0 = (float)(bf16_t)x[0], h1 = (float)(bf16_t)x[1];
        reinterpret_cast<unsigned*>(dchl)[ot] = bfpair(h0, h1); reinterpret_cast<unsigned*>(dchl + HD)[ot] = bfpair(x[0] - h0, x[1] - h1);
      }
      // a(t) and the d a tile of ctx: issued before the barrier
      float aj[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) aj[j] = (olane + 64 * j < T) ? p.a_all[((size_t)t * B + min(arow, B - 1)) * T + olane + 64 * j] : 0.f;
      bf16x8 cxv[16];
      {
        const bf16_t* r1 = cx + (size_t)min(16 * wave + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
        for (int s = 0; s < 16; ++s) cxv[s] = *reinterpret_cast<const bf16x8*>(r1 + 32 * s);
      }
      lds_barrier();
      const unsigned char* hrow = reinterpret_cast<const unsigned char*>(dchl) + 16 * q;
      for (int tile = wave; tile < ntile; tile += 4) {
        if (tile != wave) {
          const bf16_t* r2 = cx + (size_t)min(16 * tile + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
          for (int s = 0; s < 16; ++s) cxv[s] = *reinterpret_cast<const bf16x8*>(r2 + 32 * s);
        }
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int hh = 0; hh < 4; ++hh) {                           // (fragment reads in blocks of four -- eight spill beside the 304 weight registers and the ctx tile: same products, same order per accumulator)
          { Frag4 f; lds_read4(f, hrow + 256 * hh);
#pragma unroll
            for (int s = 0; s < 4; ++s) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cxv[4 * hh + s], f.v[s], a0, 0, 0, 0); }
          { Frag4 f; lds_read4(f, hrow + HD * 2 + 256 * hh);
#pragma unroll
            for (int s = 0; s < 4; ++s) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cxv[4 * hh + s], f.v[s], a1, 0, 0, 0); }
        }
        if (c16 == 0) *reinterpret_cast<f32x4*>(da + 16 * tile + 4 * q) = a0 + a1;
      }
      fetch_next();
      bf16x8 cv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + (size_t)min(4 * i + wave, T - 1) * HD + 8 * olane);
      lds_barrier();
      float dsj[4];
      {
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dsj[j] = lane + 64 * j < T ? da[lane + 64 * j] : 0.f; dot += aj[j] * dsj[j]; }
        dot = wave_reduce(dot, 0.f, [](float a, float b) { return a + b; });
#pragma unroll
        for (int j = 0; j < 4; ++j) dsj[j] = aj[j] * (dsj[j] - dot);          // SoftMax backward (LSTM.lua:139)
      }
      const float dsv_own = wave == 0 ? dsj[0] : wave == 1 ? dsj[1] : wave == 2 ? dsj[2] : dsj[3];      // d s[tid]
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
          const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dsj[ch]), 4 * i + wave));     // 0 beyond T
#pragma unroll
          for (int e = 0; e < 8; ++e) cacc[e] = fmaf(d, (float)cv[i][e], cacc[e]);
        }
      }
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8) = f32x4{cacc[0], cacc[1], cacc[2], cacc[3]};
      *reinterpret_cast<f32x4*>(part + wave * HD + lane * 8 + 4) = f32x4{cacc[4], cacc[5], cacc[6], cacc[7]};
      lds_barrier();
      float v0 = 0.f, v1 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { v0 += part[w * HD + 2 * tid]; v1 += part[w * HD + 2 * tid + 1]; }
      pst4(rvalid ? (void*)(p.dq_b + (size_t)t * slot + (size_t)arow * HD + 2 * ot) : (void*)otrash, sane(bfpair(v0, v1)), local);
      st4f(rvalid && ot < T ? (void*)(p.ds_all + ((size_t)t * B + arow) * T + ot) : (void*)otrash, dsv_own);
      asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(rvalid ? (void*)(p.dq + (size_t)t * slot + (size_t)arow * HD + 2 * ot) : (void*)otrash), "v"(f32x2{v0, v1}) : "memory");
    };
    // d z of (row, 4 units) x 4 gates of wave 0: bf16 [row][gate * 512 + unit] published, fp32 stored = 8 stores per wave
    auto dz_store = [&](int l, int rb, const f32x4 (&dz)[4]) {
      const int erow = rb + oc16; const bool eok = wave == 0 && erow < B;
      const size_t o = ((size_t)t * B + min(erow, B - 1)) * 4 * HD + 16 * member + 4 * oq;
#pragma unroll
      for (int g = 0; g < 4; ++g) pst8(eok ? (void*)(p.dzb[l] + o + g * HD) : (void*)otrash, u32x2{sane(bfpair(dz[g][0], dz[g][1])), sane(bfpair(dz[g][2], dz[g][3]))}, local);
#pragma unroll
      for (int g = 0; g < 4; ++g) st16f(eok ? (void*)(p.dz[l] + o + g * HD) : (void*)otrash, dz[g]);
    };
    // =================== B4: d h2 = d q W_a + d h2a + d h2rec; cell backward of layer 2.  stores: 8
    auto B4 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const X = xbuf(c); const int rb = row0 + RC * c;
      if constexpr (c == 0) { if (mych == 1) pend_land<3>(p.dq_b + (size_t)t * slot, HD * 2, rb, B, X, PA, olane, wave, member, local, p.err, 23, &s_dead);     // behind B3<1>
                              else pend_land<0>(p.dq_b + (size_t)t * slot, HD * 2, rb, B, X, PA, olane, wave, member, local, p.err, 23, &s_dead); }
      else pend_land<8>(p.dq_b + (size_t)t * slot, HD * 2, rb, B, X, PA, olane, wave, member, local, p.err, 23, &s_dead);                                          // behind B4<0>
      CH_STAMP(10 + c);
      Frag4 f; lds_read4(f, X + (size_t)c16 * PA + (128 * wave + 8 * q) * 2);
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, v[2];
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wat[s], f.v[s], acc[0], 0, 0, 0);
      *reinterpret_cast<f32x4*>(red + ((size_t)wave * 64 + lane) * 4) = acc[0];
      if constexpr (c == 0) { cell_issue(1, 1, 1, t, oc16, oq); pend_issue(p.dq_b + (size_t)t * slot, HD * 2, row0 + RC, B, xbuf(1), PA, olane, wave, member, local); }
      else cell_issue(0, 0, 0, t, oc16, oq);
      lds_barrier();
      // (d z2 of chain 0 left its producers at the end of B4<0>, a short phase ago: the idle waves hold their fetch back a little, wave 0 issues its own behind the cell arithmetic)
      if constexpr (c == 1) { if (wave != 0) { __builtin_amdgcn_s_sleep(14); dz_issue(p.dzb[1] + (size_t)t * B * 4 * HD, row0, xbuf(0), olane); } }
      f32x4 dz[4] = {acc[0], acc[0], acc[0], acc[0]};
      if (wave == 0) {
        v[0] = *reinterpret_cast<const f32x4*>(red + ((size_t)0 * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v[0] += *reinterpret_cast<const f32x4*>(red + ((size_t)w * 64 + lane) * 4);
        const f32x4 dh2 = v[0] + keep[(c * 2 + 0) * 64 + lane] + dh2rec[c];
        cell_bwd(c, dh2, dc2[c], dz);
        if constexpr (c == 1) dz_issue(p.dzb[1] + (size_t)t * B * 4 * HD, row0, xbuf(0), olane);
      }
      dz_store(1, rb, dz);
    };
    // =================== B5: [d h1 ; d h2rec] from d z2; cell backward of layer 1.  stores: 8
    auto B5 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const X = xbuf(c); const int rb = row0 + RC * c;
      wait_vm<8>();                                                  // behind B4<1> / B5<0>: 8 stores
      CH_STAMP(12 + c);
      f32x4 v[2], dz[4];
      zprod(wz[0], wz[1], p.dzb[1] + (size_t)t * B * 4 * HD, rb, X, v, 24, [&] {
        if constexpr (c == 0) { cell_issue(1, 1, 0, t, oc16, oq); if (t > 0) stin_issue(0, t - 1, oc16, oq, olane); }
        else { if (t > 0) stin_issue(1, t - 1, oc16, oq, olane); }
      }, [&] {
        if constexpr (c == 0) dz_issue(p.dzb[1] + (size_t)t * B * 4 * HD, row0 + RC, xbuf(1), olane);
        else dz_issue(p.dzb[0] + (size_t)t * B * 4 * HD, row0, xbuf(0), olane);
      });
      dz[0] = dz[1] = dz[2] = dz[3] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (wave == 0) {
        dh2rec[c] = v[1];
        const f32x4 dh1 = v[0] + keep[(c * 2 + 1) * 64 + lane];
        cell_bwd(c, dh1, dc1[c], dz);
      }
      dz_store(0, rb, dz);
    };
    // =================== B6: [d h1rec ; d feed] from d z1; d pre of step t-1.  stores: 2
    auto B6 = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      unsigned char* const X = xbuf(c); const int rb = row0 + RC * c;
      if constexpr (c == 0) wait_vm<8>(); else wait_vm<2>();         // behind B5<1> (8 stores) / B6<0> (2)
      CH_STAMP(14 + c);
      f32x4 v[2];
      zprod(wz[2], wz[3], p.dzb[0] + (size_t)t * B * 4 * HD, rb, X, v, 26, [&] {
        if constexpr (c == 1) { if (t > 0) pend_issue(p.dpre_b + (size_t)(t - 1) * slot, HD * 2, row0, B, xbuf(0), PA, olane, wave, member, local); }
      }, [&] {
        if constexpr (c == 0) dz_issue(p.dzb[0] + (size_t)t * B * 4 * HD, row0 + RC, xbuf(1), olane);
      });
      if (wave == 0) {
        keep[(c * 2 + 1) * 64 + lane] = v[0];                        // d h1rec
        if (t == 0) {                                                // the gradients of the initial state that leave through the last step: d h1rec, d feed
          const int erow = rb + oc16;
          if (erow < B) { const size_t o = (size_t)erow * HD + 16 * member + 4 * oq; *reinterpret_cast<f32x4*>(p.dh_rec[0] + o) = v[0]; *reinterpret_cast<f32x4*>(p.dfeed + o) = v[1]; }
        }
      }
      dpre_publish(c, t - 1, ot, true, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, v[1]);
    };

    B2(IC<0>{}); CH_STAMP(0); B2(IC<1>{}); CH_STAMP(1); if (s_dead) { dead = true; break; }
    B3(IC<0>{}); CH_STAMP(2); B3(IC<1>{}); CH_STAMP(3); if (s_dead) { dead = true; break; }
    B4(IC<0>{}); CH_STAMP(4); B4(IC<1>{}); CH_STAMP(5); if (s_dead) { dead = true; break; }
    B5(IC<0>{}); CH_STAMP(6); B5(IC<1>{}); CH_STAMP(7); if (s_dead) { dead = true; break; }
    B6(IC<0>{}); CH_STAMP(8); B6(IC<1>{}); CH_STAMP(9); if (s_dead) { dead = true; break; }
    prefill(t - 2, ot, local);
    first = false;
  }
  // d c / d h of the initial decoder state, for the encoder's backward pass (model.lua:662-690)
  if (!s_dead && wave == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int erow = row0 + RC * c + c16;
      if (erow < B) {
        const size_t o = (size_t)erow * HD + 16 * member + 4 * q;
        *reinterpret_cast<f32x4*>(p.dc_st[0] + o) = dc1[c]; *reinterpret_cast<f32x4*>(p.dc_st[1] + o) = dc2[c];
        *reinterpret_cast<f32x4*>(p.dh_rec[1] + o) = dh2rec[c];                // (d h1rec and d feed: stored by B6 of step 0)
      }
    }
  }
  wait_vm<0>();
#ifdef DC_DEBUG_STAMPS
  if (p.stamps && wid == 0 && tid == 0) { for (int k = 0; k < 16; ++k) p.stamps[k] = stamp[k]; p.err[16 + 2048 + 8] = zretry; }
#endif
}

// ---------------------------------------------------------------------------------------------
bool dec_chain_enabled() { const char* e = getenv("AOCR_NO_DEC_CHAINS"); return !(e && e[0] == '1'); }

void dec_chain_forward(hipStream_t s, const DecClFwdArgs& a0, bool greedy_decode) {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_FWD_LDS);
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_FWD_LDS);
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_DEC_LDS);
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_DEC_LDS);
  const bool res = a0.T <= 64 && !getenv("AOCR_CH_NO_RES");
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClFwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    { const char* e = getenv("AOCR_DC_STAMPS"); a.stamps = (e && e[0] != 'b') ? a.xtab + (size_t)groups * NM : nullptr; }      // ("beam": only the beam-search launches)
    const dim3 grid(8 * NM * ((a.ngroups + 7) / 8));
    if (greedy_decode) {
      a.pgroups = groups; a.tokx = reinterpret_cast<unsigned*>(a.pbuf + (size_t)2 * groups * 32 * 32 * 40); a.no_early = getenv("AOCR_NO_DEC_EARLY") != nullptr;
      if (res) hipLaunchKernelGGL((dec_ch_fwd_kernel<true, true>), grid, dim3(256), (size_t)CH_DEC_LDS, s, a);
      else hipLaunchKernelGGL((dec_ch_fwd_kernel<true, false>), grid, dim3(256), (size_t)CH_DEC_LDS, s, a);
    } else if (res) hipLaunchKernelGGL((dec_ch_fwd_kernel<false, true>), grid, dim3(256), (size_t)CH_FWD_LDS, s, a);
    else hipLaunchKernelGGL((dec_ch_fwd_kernel<false, false>), grid, dim3(256), (size_t)CH_FWD_LDS, s, a);
  }
}

// ---- beam search on the chain kernel.  Groups hold 2 * (16 / k) images; a launch runs at most one group per XCD, and every launch needs its own
// epoch (the exchange buffers are indexed by the group WITHIN the launch): a0.epoch is the first of dec_chain_beam_passes() consecutive ones.
static int chain_per_pass() {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return std::max(8, cus / (8 * NM) * 8);
}
static int beam_groups(int B, int k) { const int ipg = NCH * (RC / k); return (B + ipg - 1) / ipg; }
// groups of one launch: one per XCD at most, and no more than fit four rolling step slots (32 rows each per group) into the [L + 1][B][Hd] / [L][B][2 Hd]
// buffers behind the B rows of the initial state: 4 * 32 g <= L B
static int beam_pass_groups(int B, int L, int k) { return (int)std::min<long>(std::min(chain_per_pass(), beam_groups(B, k)), (long)L * B / 128); }
int dec_chain_beam_passes(int B, int L, int k) { const int g = beam_pass_groups(B, L, k); return g < 1 ? 0 : (beam_groups(B, k) + g - 1) / g; }
int dec_chain_beam_group_cap() { return chain_per_pass(); }
bool dec_chain_beam_supported(int B, int L, int k, int V) {
  if (k < 2 || k > 8 || V > 40 || B < 1 || getenv("AOCR_NO_DEC_CHAINS_BEAM") || !dec_chain_enabled()) return false;
  return beam_pass_groups(B, L, k) >= 1;
}
void dec_chain_beam_forward(hipStream_t s, const DecClFwdArgs& a0) {
  const int groups = beam_groups(a0.B, a0.beam), per_pass = beam_pass_groups(a0.B, a0.L, a0.beam), cap = chain_per_pass();
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CH_DEC_LDS + CH_BEAM_EXTRA));
  (void)hipFuncSetAttribute((const void*)dec_ch_fwd_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CH_DEC_LDS + CH_BEAM_EXTRA));
  const bool res = a0.T <= 64 && !getenv("AOCR_CH_NO_RES");
  unsigned epoch = a0.epoch;
  for (int g0 = 0; g0 < groups; g0 += per_pass, ++epoch) {
    DecClFwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    a.epoch = epoch; a.stamps = getenv("AOCR_DC_STAMPS") ? a.xtab + (size_t)cap * NM : nullptr; a.no_early = 1;
    a.rows_slot = 32 * per_pass;
    a.pgroups = cap; a.tokx = reinterpret_cast<unsigned*>(a.pbuf + (size_t)2 * cap * 32 * 32 * 40);
    const dim3 grid(8 * NM * ((a.ngroups + 7) / 8));
    if (res) hipLaunchKernelGGL((dec_ch_fwd_kernel<true, true, true>), grid, dim3(256), (size_t)(CH_DEC_LDS + CH_BEAM_EXTRA), s, a);
    else hipLaunchKernelGGL((dec_ch_fwd_kernel<true, false, true>), grid, dim3(256), (size_t)(CH_DEC_LDS + CH_BEAM_EXTRA), s, a);
  }
}

void dec_chain_backward(hipStream_t s, const DecClBwdArgs& a0) {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  (void)hipFuncSetAttribute((const void*)dec_ch_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CH_BWD_LDS);
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClBwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    a.stamps = getenv("AOCR_DC_STAMPS") ? a.xtab + (size_t)groups * NM + 16 : nullptr;
    hipLaunchKernelGGL(dec_ch_bwd_kernel, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), (size_t)CH_BWD_LDS, s, a);
  }
}

}  // namespace aocr
