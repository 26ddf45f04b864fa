// rnn_cluster.hip -- the BiLSTM encoder recurrence (model.lua:291-316 forward, :662-690 BPTT; LSTM.lua:79-105 cell) on CLUSTERS of
// compute units, bf16 operands / fp32 accumulate.
//
// rnn_seq.hip gives one workgroup 16 batch rows and re-streams the whole recurrent matrix through it every step (32 of 256 CUs busy
// at C3, 320 KB of weights per step and workgroup).  Here a GROUP of G = He/64 workgroups (one per CU, dealt to one XCD) shares a
// block of 16*RT rows: every workgroup owns 64 hidden units -- 256 of the 4He gate columns -- and keeps its slice of W_h2h RESIDENT
// IN REGISTERS for the whole sequence (He/2 VGPRs per lane: the B fragments of its MFMAs, loaded once); no weight byte moves after the
// prologue, at He = 256 and at He = 512 alike.  What moves per step is the state:
//   forward   h(t) slices, ALL-GATHERED: each wave publishes its 16 rows x 16 units as 8-byte {2 x bf16, tag} granules (one sc1 store
//             per granule, tag = launch epoch | step), every wave of the group polls the granules it needs with sc1 loads until the
//             tags match -- no flags, no fences, no barriers, no LDS (MI355X_MICROARCH.md: data-tagged granules, handoff-1to1);
//   backward  d h = d z . W_h2h has K = 4He and N = He, so the roles flip: each workgroup multiplies ITS d z (own 256 gate columns,
//             through LDS) by its 256 rows of W_h2h and the partial sums are REDUCE-SCATTERED -- fp32 granules in the accumulator's
//             own lane layout, so sender and receiver touch 32 contiguous bytes per lane.
// Buffers alternate with the step parity; a slot is only rewritten two steps later, after every reader has published the step in
// between (which it can only do having consumed the slot).  Polls are bounded: a timeout sets *err and the kernel runs out.
// Placement matters for speed only (blocks b and b+8 share an XCD under round-robin dispatch), never for correctness.
#include "ops.h"
#include <algorithm>
#include <cstdlib>
#include <cstdio>

namespace aocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

namespace {
constexpr int CL_SPIN_LIMIT = 1 << 18;

__device__ __forceinline__ u64 ld_granule(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_granule(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned bf16_bits(float x) { bf16_t h = (bf16_t)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
}  // namespace

// ---------------------------------------------------------------------------------------------
// VMEM discipline.  The group's critical path is  granule store -> L2 -> poll.  Everything else a step moves (state slots, saved
// gates, the next step's inputs) is bulk traffic whose completion must NOT be waited for on that path: vmcnt counts in issue order, so
// the polls are issued FIRST, the bulk stores / prefetch loads of the neighbouring steps BEHIND them, and the wait for the polls is
// an explicit s_waitcnt vmcnt(N) with N = the number of bulk instructions issued in between.  That count must be static: the polls are
// inline asm (the compiler's own wait insertion would drain the queue), invalid rows store to a trash slot instead of branching.
// ---------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void poll16(u32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); }
__device__ __forceinline__ void pin(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }
// bulk traffic as inline asm too: its instruction count between a batch of polls and their wait must be exact, and the compiler must not
// insert a wait of its own in front of a use of a prefetched value (it would drain the stores issued since)
__device__ __forceinline__ void ld4(float& v, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld16(f32x4& v, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void st4(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st2(bf16_t* p, unsigned bits) { asm volatile("global_store_short %0, %1, off" ::"v"(p), "v"(bits) : "memory"); }
// NOTE the trailing s_nop: a store of more than 8 bytes reads its data registers for a few cycles after issue, and the compiler's hazard
// recognizer does not look inside inline asm -- without the wait states the next VALU write to one of those registers (the register
// allocator reuses them at once) corrupts the data of the later lanes (seen: rows 12-15 of a tile).
__device__ __forceinline__ void st16(void* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8(void* p, unsigned a, unsigned b) { typedef unsigned u32x2 __attribute__((ext_vector_type(2))); const u32x2 v = {a, b}; asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
// Prefetch WITHOUT a register destination (LDS-DMA): lane l of the wave gets its 16 bytes at lds_base + 16 l.  An inline-asm load whose
// result must stay untouched in a VGPR across an MFMA phase is not safe -- under register pressure the allocator splits the live range
// (copies the not-yet-landed register to an AGPR) -- so everything prefetched across a phase goes through LDS and is read back behind
// the wait that covers it (vmcnt counts LDS-DMA like any other load).
// Inline asm (M0 written in the statement that reads it), NOT __builtin_amdgcn_global_load_lds: hipcc tracks the builtin as an LDS write and drains
// vmcnt to 0 at the next workgroup barrier -- which also waits for the acknowledgement of the granule stores issued since.  The waits that cover
// these loads are the explicit counted ones.
__device__ __forceinline__ void cl_dma16(const void* g, unsigned char* lds_base) {
  const unsigned a = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) void*)lds_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(a) : "memory");
}
// Eight B fragments (16 bytes each, 64 bytes apart) in ONE asm statement: issued back to back, waited for once (dec_chain.hip's lds_read8) -- hipcc otherwise emits
// read / wait / MFMAs per k-step (C5, G = 8: 2.4 k cycles for the 64 MFMAs of a step instead of ~1.2 k)
struct ClFrag8 { bf16x8 v[8]; };
__device__ __forceinline__ void cl_read8(ClFrag8& f, const unsigned char* p) {
  const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\tds_read_b128 %3, %8 offset:192\n\t"
               "ds_read_b128 %4, %8 offset:256\n\tds_read_b128 %5, %8 offset:320\n\tds_read_b128 %6, %8 offset:384\n\tds_read_b128 %7, %8 offset:448\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(f.v[0]), "=&v"(f.v[1]), "=&v"(f.v[2]), "=&v"(f.v[3]), "=&v"(f.v[4]), "=&v"(f.v[5]), "=&v"(f.v[6]), "=&v"(f.v[7]) : "v"(a) : "memory");
}
// workgroup barrier whose fences cover LDS only (no wait for outstanding global stores)
__device__ __forceinline__ void cl_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// Two granules to a consumer.  `local` (wave-uniform): every member of the group runs on ONE XCD (checked at kernel start), so a plain
// store -- which stays in that XCD's L2 -- is visible to the members' sc1 (L1-bypassing) polls without the trip through the fabric that
// a write-through store costs (MI355X_MICROARCH.md: plain stores KEEP the line in the XCD's L2, sc1 stores DROP it).
__device__ __forceinline__ void st_granules(void* p, u32x4 v, bool local) {
  if (local) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else st16_sc1(p, v);
}
// Are all G members of this group on one XCD?  Each member publishes its XCC id (tagged with the launch epoch), reads the others'.
template <int G> __device__ __forceinline__ bool group_is_local(u64* tab, int member, unsigned epoch, int* err) {
  __shared__ int s_local;
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    st_granule(tab + member, ((u64)epoch << 32) | (u64)(xcc + 1u));
    int same = 1;
    for (int m = 0; m < G; ++m) {
      u64 v; int spins = 0;
      while ((unsigned)((v = ld_granule(tab + m)) >> 32) != epoch) { if (++spins > CL_SPIN_LIMIT) { atomicExch(err, 3); same = 0; break; } __builtin_amdgcn_s_sleep(2); }
      if ((unsigned)v != xcc + 1u) same = 0;
    }
    s_local = same;
  }
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(s_local) != 0;
}   // two granules, write-through

// Both kernels compute the TRANSPOSED products (weights as the A operand, the state as B): an accumulator then holds FOUR CONSECUTIVE
// UNITS of ONE batch row per lane, so every global access of the epilogue is 8 or 16 contiguous bytes per lane (7 stores and 4 loads
// per 16-row tile and step instead of 16 + 16) and a lane's h values are one 16-byte pair of granules.  With the polls that is ~28
// vector-memory instructions per wave and step; the first layout issued 56, and the wave (at most 63 in flight) stalled on its
// own store acknowledgements.
//
#ifdef CL_TIMING_FINE
#define CL_FINE(k) do { const u64 now_ = __builtin_readcyclecounter(); fin_[k] += now_ - cprev_; cprev_ = now_; } while (0)
#else
#define CL_FINE(k) do { } while (0)
#endif
#ifdef DC_DEBUG_STAMPS
#define CL_STAMP(k) do { const u64 now_ = __builtin_readcyclecounter(); cst_[k] += now_ - cprev_; cprev_ = now_; } while (0)
#else
#define CL_STAMP(k) do { } while (0)
#endif
// forward.  Saved gates layout of the cluster kernels: [T][B][He][4] (16 bytes per cell).
template <int G, int RT, bool CTX>
__global__ __launch_bounds__(256, 1) void enc_cl_fwd_kernel(EncClFwdArgs p) {
  constexpr int He = 64 * G, KS = 2 * G, R = 16 * RT;
  constexpr int NPW = (2 * G + 3) / 4;                           // (member, k-step) pieces polled per wave
  constexpr int HP = He * 2 + 16;                                // LDS pitch of the gathered h(t-1): [R][He] bf16, padded
  constexpr int NBULK = RT * (CTX ? 7 : 6);                      // store instructions issued behind the polls (inline asm: exact)
  extern __shared__ __attribute__((aligned(16))) unsigned char hbuf[];        // [2][R][HP], then the zx staging [4 waves][RT * 4 pieces][1 KiB]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % G, gl = (i8 / G) * 8 + xcd;           // the G members of a group: blocks 8 apart = one XCD (speed only)
  if (gl >= p.ngid) return;
  const int gid = p.gid0 + gl;
  const int dir = gid / p.groups, group = gid - dir * p.groups;
  // A COPY whose fields are pinned in scalar registers: through the reference hipcc re-loaded the pointers from the kernarg segment inside the step loop
  // (s_load_dwordx2 + s_waitcnt lgkmcnt(0) in front of every output store: 3.4 k of the 7.2 k cycles of a step, tools/debug/enc_stamp.py)
  EncSeqDir d = p.d[dir];
  asm volatile("" : "+s"(d.zx), "+s"(d.hs), "+s"(d.cs), "+s"(d.hsb), "+s"(d.gates), "+s"(d.ctx), "+s"(d.reverse));
  // RH = batch rows a 16-column tile carries (p.rh: 16, or 8 = HALF tiles: twice the groups on twice the compute units, half the output bytes per CU
  // and step -- the output stores are what bounds a step, see store_outputs; columns RH..15 repeat column RH-1 and are never stored or published)
  const int RH = p.rh == 8 ? 8 : 16;
  const int B = p.B, T = p.T, row0 = group * (RH * RT);
  const int cv = min(c16, RH - 1);                              // the tile column this lane READS (its own, or the last valid one)
  const bool cok = c16 < RH;                                    // ... and whether it owns one
  const int u0 = 64 * member + 16 * wave + 4 * q;               // this lane's four hidden units u0 .. u0+3 (epilogue), batch row c16
  unsigned char* const zxl = hbuf + 2 * R * HP + wave * (RT * 4 * 1024);

  bf16x8 wres[4][KS];                                            // resident A fragments: gate g, k-step s; A row = unit 16 wave + c16
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int s = 0; s < KS; ++s)
      wres[g][s] = *reinterpret_cast<const bf16x8*>(d.w + (size_t)(g * He + 64 * member + 16 * wave + c16) * He + 32 * s + 8 * q);

  f32x4 cst[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) cst[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  u64* const xg = p.xbuf + (size_t)(gid + p.gslot) * 2 * G * R * 32;         // [parity][member][row][32 granules]
  float* const trash = reinterpret_cast<float*>(p.err + 16) + (threadIdx.x & 255) * 4;   // rows >= B store here
  const bool local = group_is_local<G>(p.xtab + (size_t)(gid + p.gslot) * 8, member, p.epoch, p.err) && !p.force_remote;
  // This launch runs iterations it0 .. it1-1 of the T (a CHUNK of the sequence: model.hip runs the layers of a stacked encoder as a
  // wavefront of chunks on separate streams).  A chunk that does not start the sequence takes c and h of the iteration before from the
  // state slots the previous chunk's launch wrote.
  const int it0 = p.it0, it1 = p.it1 > 0 ? p.it1 : T;

  auto dma_zx = [&](int t) {                                     // RT * 4 LDS-DMA loads: the input part of step t for this lane's cells
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = min(row0 + RH * rt + cv, B - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) cl_dma16(d.zx + ((size_t)t * B + row) * 4 * He + g * He + u0, zxl + (rt * 4 + g) * 1024);
    }
  };
  dma_zx(d.reverse ? T - 1 - it0 : it0);
  if (it0 > 0) {
    const int tp = d.reverse ? T - it0 : it0 - 1;               // the step of iteration it0 - 1: its state is in slot tp + 1
    unsigned char* const hb = hbuf + (size_t)(it0 & 1) * R * HP;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { const int row = min(row0 + RH * rt + cv, B - 1); cst[rt] = *reinterpret_cast<const f32x4*>(d.cs + ((size_t)(tp + 1) * B + row) * He + u0); }
    for (int x = threadIdx.x; x < R * (He / 8); x += 256) {
      const int rr = x / (He / 8), k8 = x - rr * (He / 8), row = min(row0 + RH * (rr >> 4) + min(rr & 15, RH - 1), B - 1);
      *reinterpret_cast<u32x4*>(hb + (size_t)rr * HP + k8 * 16) = *reinterpret_cast<const u32x4*>(d.hsb + ((size_t)(tp + 1) * B + row) * He + k8 * 8);
    }
    __syncthreads();
  }
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0) the compiler can see: no wait of its own for the prologue loads inside the loop
  f32x4 gat[RT][4], hh[RT], cc[RT];                              // outputs of the step before ([unit i] = {in, forget, out, g}), stored one step late
  [[maybe_unused]] u64 fin_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cprev_ = 0;
  // What a step costs (tools/debug/enc_stamp.py, stamps build, C3: 7.2 k cycles): these stores are 2.9 k of it -- 26 KB per CU and step through a write path
  // that sustains ~16 bytes per clock and CU (1.7 k cycles even when every store goes to one fixed 4 KB), + ~1.2 k for the 64-byte-per-row pieces of the
  // real layouts.  They overlap the exchange (polls answered after ~2.4 k cycles), so the net cost is ~1.7 k cycles per step.  Measured without gain:
  // the gates behind the barrier (the stall moves to the next step's polls), lane-contiguous stores (-15 %), touching the pages one phase ahead (worse).
  // The lever that is left is bytes per CU: 32 units per member (twice the CUs per group; the idle half of the chip at C3, 15/16 of it at C5).
  auto store_outputs = [&](int t, int part) {                    // part 0: c, h, context = RT * (2 or 3) store instructions; part 1: the saved gates = RT * 4; none conditional
    const size_t so = (size_t)(t + 1) * B * He;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = row0 + RH * rt + c16;
      const bool ok = row < B;
      const size_t o = so + (size_t)row * He + u0;
      if (!cok) continue;                                        // (an exec mask, not a branch around the instructions: every wave has owners, the store count stays static)
      if (part == 0) {
      st16(ok ? d.cs + o : trash, cc[rt]);
      CL_FINE(0);
      const unsigned h01 = bf16_bits(hh[rt][0]) | (bf16_bits(hh[rt][1]) << 16), h23 = bf16_bits(hh[rt][2]) | (bf16_bits(hh[rt][3]) << 16);
      st8(ok ? reinterpret_cast<void*>(d.hsb + o) : reinterpret_cast<void*>(trash), h01, h23);
      CL_FINE(1);
      if (CTX) st16(ok ? d.ctx + ((size_t)row * T + t) * p.Hd + u0 : trash, hh[rt]);
      CL_FINE(2);
      } else {
      float* gp = ok ? d.gates + (((size_t)t * B + row) * He + u0) * 4 : trash;
#pragma unroll
      for (int i = 0; i < 4; ++i) { st16(ok ? gp + 4 * i : trash, gat[rt][i]); CL_FINE(3 + i); }
      }
    }
  };
  bool dead = false;
  __shared__ int s_dead;
  if (threadIdx.x == 0) s_dead = 0;
  __syncthreads();

  [[maybe_unused]] u64 cst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; cprev_ = __builtin_readcyclecounter();
  for (int it = it0; it < it1 && !dead; ++it) {
    const int t = d.reverse ? T - 1 - it : it;
    const int tprev = d.reverse ? t + 1 : t - 1;
    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned tag = p.epoch * 4096u + (unsigned)it;
    const u64* xp = xg + (size_t)((it - 1) & 1) * G * R * 32 + cv * 32 + 4 * q;
    if (it > it0) {
      // ---- all-gather of h(t-1), shared by the four waves: the 2G (member, k-step) pieces are dealt to the waves (piece x = wave + 4 j),
      // each wave polls its pieces (2 RT loads of 16 bytes per lane and piece), drops the bf16 payload into LDS as the B operand
      // [row][k], and after ONE barrier every wave reads its fragments for all of K.  LDS alternates with the step parity.
      unsigned char* const hb = hbuf + (size_t)(it & 1) * R * HP;
      u32x4 gr[NPW][RT][2];
      auto issue = [&]() {
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
          const int x = (wave + 4 * j) % (2 * G);                // straight-line on purpose (G = 1: waves 2, 3 repeat pieces 0, 1): a branch
                                                                 // around an asm load makes the compiler copy its not-yet-landed result
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) poll16(gr[j][rt][hf], xp + (size_t)(x >> 1) * R * 32 + rt * 16 * 32 + (x & 1) * 16 + hf * 2);
        }
      };
      auto settle = [&]() {
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) pin(gr[j][rt][hf]);
      };
      CL_STAMP(0);
      issue();
      CL_STAMP(7);
      store_outputs(tprev, 0); store_outputs(tprev, 1);                                      // behind the polls: the outputs of the step before (NBULK store instructions)
      CL_STAMP(1);
      wait_vm<NBULK>();                                          // the polls and everything older (this step's zx in LDS) have landed
      settle();
      CL_STAMP(2);
      int spins = 0;
      while (true) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) ok = ok && gr[j][rt][hf][1] == tag && gr[j][rt][hf][3] == tag;
        if (__all(ok)) break;
        if (++spins > CL_SPIN_LIMIT) {
          dead = true;
          if (lane == 0 && atomicExch(p.err, 1) == 0) { p.err[1] = it; p.err[2] = wave; p.err[3] = member; p.err[4] = gid; }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        issue(); wait_vm<0>(); settle();
      }
      CL_STAMP(3);
#pragma unroll
      for (int j = 0; j < NPW; ++j) {
        const int x = (wave + 4 * j) % (2 * G);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const u32x4 raw = {gr[j][rt][0][0], gr[j][rt][0][2], gr[j][rt][1][0], gr[j][rt][1][2]};       // units 32 x + 8 q .. +7 of batch row c16
          *reinterpret_cast<u32x4*>(hb + (size_t)(16 * rt + c16) * HP + (32 * x + 8 * q) * 2) = raw;
        }
      }
      // ONE barrier, LDS-only fences: __syncthreads_or() expands to three more barriers, a scalar load of the dispatch packet and -- through its
      // memory fences -- s_waitcnt vmcnt(0), i.e. the acknowledgement of this step's young stores, on the path every step takes
      if (dead && lane == 0) s_dead = 1;                         // a timeout anywhere in the workgroup stops all of it (plain LDS accesses: a volatile generic access is a FLAT instruction, counted by vmcnt too)
      cl_barrier();
      dead = s_dead != 0;
      if (dead) break;
      CL_STAMP(4);
    }
    if (it > 0) {                                                // (the first iteration of a later chunk: h(t-1) was put into LDS by the prologue)
      const unsigned char* const hb = hbuf + (size_t)(it & 1) * R * HP;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int s8 = 0; s8 < KS; s8 += 8) {
          if constexpr (KS % 8 == 0) {
            ClFrag8 hv; cl_read8(hv, hb + (size_t)(16 * rt + c16) * HP + (32 * s8 + 8 * q) * 2);                             // B operand: column = batch row c16
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wres[g][s8 + s2], hv.v[s2], acc[rt][g], 0, 0, 0);
          } else {
#pragma unroll
            for (int s2 = s8; s2 < KS; ++s2) {
              const bf16x8 hv = *reinterpret_cast<const bf16x8*>(hb + (size_t)(16 * rt + c16) * HP + (32 * s2 + 8 * q) * 2);
#pragma unroll
              for (int g = 0; g < 4; ++g) acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wres[g][s2], hv, acc[rt][g], 0, 0, 0);
            }
          }
        }
    }
    if (dead) break;
    CL_STAMP(5);
    f32x4 zx[RT][4];                                             // this step's input part, prefetched into LDS one step ago
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) zx[rt][g] = *reinterpret_cast<const f32x4*>(zxl + (rt * 4 + g) * 1024 + lane * 16);
    // ---- gate math (acc[rt][g][i]: unit u0 + i, batch row c16) and the exchange store: one 16-byte pair of granules per lane and tile
    const unsigned tagn = p.epoch * 4096u + (unsigned)(it + 1);
    u64* const xw = xg + ((size_t)(it & 1) * G + member) * R * 32 + 8 * wave + 2 * q;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ig = sigmoidf_(acc[rt][0][i] + zx[rt][0][i]), fg = sigmoidf_(acc[rt][1][i] + zx[rt][1][i]);
        const float og = sigmoidf_(acc[rt][2][i] + zx[rt][2][i]), gg = tanhf_(acc[rt][3][i] + zx[rt][3][i]);
        const float cn = fg * cst[rt][i] + ig * gg;
        cst[rt][i] = cn; cc[rt][i] = cn; hh[rt][i] = og * tanhf_(cn); gat[rt][i] = f32x4{ig, fg, og, gg};
      }
      if (it + 1 < it1) {
        const u32x4 gv = {bf16_bits(hh[rt][0]) | (bf16_bits(hh[rt][1]) << 16), tagn, bf16_bits(hh[rt][2]) | (bf16_bits(hh[rt][3]) << 16), tagn};
        if (cok) st_granules(xw + (size_t)(16 * rt + c16) * 32, gv, local);
      }
    }
    dma_zx(it + 1 < it1 ? (d.reverse ? t - 1 : t + 1) : t);        // the next step's input part: issued behind the granules, older than the next polls
    CL_STAMP(6);
  }
  wait_vm<0>();
#ifdef DC_DEBUG_STAMPS
  if (wid == 0 && threadIdx.x == 0) for (int k = 0; k < 8; ++k) p.err[16 + 2048 + 1000 + k] = (int)(cst_[k] >> 4);
#ifdef CL_TIMING_FINE
  if (wid == 0 && threadIdx.x == 0) for (int k = 0; k < 8; ++k) p.err[16 + 2048 + 1010 + k] = (int)(fin_[k] >> 4);
#endif
#endif
  if (!dead) {                                                   // the last step's outputs + its fp32 h (the decoder's initial state reads it)
    const int tl = d.reverse ? T - it1 : it1 - 1;
    store_outputs(tl, 0); store_outputs(tl, 1);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) { const int row = row0 + RH * rt + c16; if (cok && row < B) *reinterpret_cast<f32x4*>(d.hs + ((size_t)(tl + 1) * B + row) * He + u0) = hh[rt]; }
  }
}

// ---------------------------------------------------------------------------------------------
// backward (BPTT) of one layer.  Per step: MFMA on the own d z of the step before (LDS) -> partial tiles sent (granules) / kept (LDS)
// -> barrier -> polls issued, behind them the bf16 d z stores of the step before -> partial sums -> EpGatesBwd -> d z(t) to LDS ->
// barrier.  The bias gradient (sum of d z over rows and steps) is kept in registers and added to both bias gradients at the end: no
// fp32 d z leaves the kernel.  Transposed tiles as in the forward kernel: lane = batch row c16, units u0 .. u0+3.
// ---------------------------------------------------------------------------------------------
template <int G, int RT>
__global__ __launch_bounds__(256, 1) void enc_cl_bwd_kernel(EncClBwdArgs p) {
  constexpr int He = 64 * G, KG = 4 * He, R = 16 * RT, AP = 256 * 2 + 16;     // B operand: [R][256 own gate columns] bf16, padded pitch
  constexpr int NBULK = RT * 4;                                 // store instructions issued behind the first batch of polls (4 x 8 bytes of bf16 d z per tile)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const abuf = lds;                                            // [2][R][AP]
  f32x4* const own = reinterpret_cast<f32x4*>(lds + 2 * R * AP);              // [4 tiles][RT][64 lanes]
  unsigned char* const prel = lds + 2 * R * AP + 4 * RT * 64 * 16 + (threadIdx.x >> 6) * (RT * 7 * 1024);   // epilogue-input staging [4 waves][RT * 7 pieces][1 KiB]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % G, gl = (i8 / G) * 8 + xcd;
  if (gl >= p.ngid) return;
  const int gid = p.gid0 + gl;
  const int dir = gid / p.groups, group = gid - dir * p.groups;
  EncSeqBwdDir d = p.d[dir];                                     // pinned copy: see the forward kernel
  asm volatile("" : "+s"(d.dh1), "+s"(d.dh1_row), "+s"(d.dh1_t), "+s"(d.dh2), "+s"(d.dh2_row), "+s"(d.dc), "+s"(d.gates), "+s"(d.cs), "+s"(d.dz), "+s"(d.dzb), "+s"(d.forward_dir));
  const int RH = p.rh == 8 ? 8 : 16;                            // batch rows per 16-column tile (8: half tiles, see the forward kernel)
  const int B = p.B, T = p.T, row0 = group * (RH * RT);
  const int cv = min(c16, RH - 1); const bool cok = c16 < RH;   // the column this lane reads; whether it owns one
  const int lane_v = (lane & 48) | cv;                          // ... as a lane index (granule slots of the columns nobody owns are never written)
  const int ul = 16 * wave + 4 * q, u0 = 64 * member + ul;      // epilogue: this lane's four hidden units (local / global), batch row c16
  float* const trash = reinterpret_cast<float*>(p.err + 16) + (threadIdx.x & 255) * 4;

  // resident A fragments: this wave's G output tiles -- tile 4 j + wave, i.e. ONE tile for every destination member j (A row = unit 16 (4 j + wave) + c16), so every
  // wave sends G - 1 tiles and keeps one (round 5; with tiles wave G .. wave G + G - 1 the wave whose tiles were the workgroup's own sent nothing and waited at the
  // barrier for the three that sent four each) --, K = the workgroup's own 256 gate columns
  // k = gate * 64 + local unit  <->  column gate * He + 64 member + local unit of W^T [He][4He]
  bf16x8 wres[G][8];
#pragma unroll
  for (int j = 0; j < G; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      wres[j][s] = *reinterpret_cast<const bf16x8*>(d.wt + (size_t)(16 * (4 * j + wave) + c16) * KG + (s >> 1) * He + 64 * member + 32 * (s & 1) + 8 * q);

  f32x4 dcr[RT]; float dbs[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) dbs[g][i] = 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int row = min(row0 + RH * rt + cv, B - 1);
    dcr[rt] = (d.dc_in && p.it0 == 0) ? *reinterpret_cast<const f32x4*>(d.dc_in + (size_t)row * d.dc_in_row + u0) : *reinterpret_cast<const f32x4*>(d.dc + (size_t)row * He + u0);
  }

  struct Pre { f32x4 g[RT][4], cc[RT], cp[RT], dh[RT]; };
  auto prefetch = [&](int it) {                                  // RT * 7 LDS-DMA loads: saved gates, c(t), c(t-1), d h from above
    const int t = d.forward_dir ? T - 1 - it : it, prev = d.forward_dir ? t : t + 2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = min(row0 + RH * rt + cv, B - 1);
      unsigned char* pl = prel + rt * 7 * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) cl_dma16(d.gates + (((size_t)t * B + row) * He + u0 + i) * 4, pl + i * 1024);
      cl_dma16(d.cs + ((size_t)(t + 1) * B + row) * He + u0, pl + 4 * 1024);
      cl_dma16(d.cs + ((size_t)prev * B + row) * He + u0, pl + 5 * 1024);
      cl_dma16(d.dh1 + (size_t)row * d.dh1_row + (size_t)t * d.dh1_t + u0, pl + 6 * 1024);
    }
  };
  auto read_pre = [&](Pre& x) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const unsigned char* pl = prel + rt * 7 * 1024 + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) x.g[rt][i] = *reinterpret_cast<const f32x4*>(pl + i * 1024);
      x.cc[rt] = *reinterpret_cast<const f32x4*>(pl + 4 * 1024); x.cp[rt] = *reinterpret_cast<const f32x4*>(pl + 5 * 1024);
      x.dh[rt] = *reinterpret_cast<const f32x4*>(pl + 6 * 1024);
    }
  };
  u64* const pb = p.pbuf + (size_t)(gid + p.gslot) * 2 * G * G * 4 * RT * 256;            // [parity][dest][src][tile][rt][half][lane][2 granules]
  const bool local = group_is_local<G>(p.xtab + (size_t)(gid + p.gslot) * 8, member, p.epoch, p.err) && !p.force_remote;
  const int it0 = p.it0, it1 = p.it1 > 0 ? p.it1 : T;          // this launch's chunk of the T iterations (see the forward kernel)
  f32x4 zprev[RT][4];                                            // d z of the step before ([gate][unit i]): stored to HBM one step late
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int g = 0; g < 4; ++g) zprev[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto store_dz = [&](int t) {                                   // RT * 4 store instructions, none conditional
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = row0 + RH * rt + c16;
      const bool ok = row < B;
      if (!cok) continue;                                        // (exec mask: the store count stays static)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const unsigned z01 = bf16_bits(zprev[rt][g][0]) | (bf16_bits(zprev[rt][g][1]) << 16), z23 = bf16_bits(zprev[rt][g][2]) | (bf16_bits(zprev[rt][g][3]) << 16);
        st8(ok ? reinterpret_cast<void*>(d.dzb + ((size_t)t * B + row) * KG + g * He + u0) : reinterpret_cast<void*>(trash), z01, z23);
      }
    }
  };
  bool dead = false;
  if (it0 > 0) {
    // a later chunk: d z of the iteration before (written as bf16 by the previous chunk's launch) is this chunk's first B operand, and
    // is what the first iteration re-stores "one step late" (bf16 -> float -> bf16 is exact)
    const int tp = d.forward_dir ? T - it0 : it0 - 1;
    unsigned char* const ab = abuf + (size_t)((it0 - 1) & 1) * R * AP;
    for (int x = tid; x < R * 32; x += 256) {                    // 16 B pieces: [row][gate][8 pieces of 8 units]
      const int rr = x >> 5, g = (x >> 3) & 3, k8 = x & 7, row = min(row0 + RH * (rr >> 4) + min(rr & 15, RH - 1), B - 1);
      *reinterpret_cast<u32x4*>(ab + (size_t)rr * AP + (g * 64 + k8 * 8) * 2) = *reinterpret_cast<const u32x4*>(d.dzb + ((size_t)tp * B + row) * KG + g * He + 64 * member + k8 * 8);
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const int row = min(row0 + RH * rt + cv, B - 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 zz = *reinterpret_cast<const u32x2*>(d.dzb + ((size_t)tp * B + row) * KG + g * He + u0);
        zprev[rt][g] = f32x4{__uint_as_float(zz[0] << 16), __uint_as_float(zz[0] & 0xFFFF0000u), __uint_as_float(zz[1] << 16), __uint_as_float(zz[1] & 0xFFFF0000u)};
      }
    }
    __syncthreads();
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0) the compiler can see: no wait of its own for the prologue loads inside the loop

  [[maybe_unused]] u64 cst_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cprev_ = __builtin_readcyclecounter();
  for (int it = it0; it < it1; ++it) {
    const int t = d.forward_dir ? T - 1 - it : it;
    const int tprev = d.forward_dir ? t + 1 : t - 1;
    const unsigned tag = p.epoch * 4096u + (unsigned)it;
    const int par = it & 1;
    CL_STAMP(0);
    prefetch(it);                                                // this step's epilogue inputs -> LDS: older than the polls, landed when they have
    if (it > 0 && !dead) {
      // ---- partial d h (transposed: units x rows) for ALL units from this workgroup's own d z of the step before
      f32x4 acc[RT][G];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < G; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* ab = abuf + (size_t)((it - 1) & 1) * R * AP;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        ClFrag8 zb; cl_read8(zb, ab + (size_t)(16 * rt + c16) * AP + (8 * q) * 2);                                           // column = batch row c16
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
          for (int j = 0; j < G; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wres[j][s], zb.v[s], acc[rt][j], 0, 0, 0);
      }
      CL_STAMP(1);
      // ---- reduce-scatter: tile 4 j + wave belongs to member j, and to its wave `wave`
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int dm = j, e = wave;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if (dm == member) own[(e * RT + rt) * 64 + lane] = acc[rt][j];
          else {
            u64* dst = pb + ((((size_t)(par * G + dm) * G + member) * 4 + e) * RT + rt) * 256 + lane * 2;      // tile = [half][lane][2 granules]: each store instruction writes 1 KB contiguous
            const u32x4 g0 = {__float_as_uint(acc[rt][j][0]), tag, __float_as_uint(acc[rt][j][1]), tag};
            const u32x4 g1 = {__float_as_uint(acc[rt][j][2]), tag, __float_as_uint(acc[rt][j][3]), tag};
            if (cok) { st_granules(dst, g0, local); st_granules(dst + 128, g1, local); }
          }
        }
      }
    }
    CL_STAMP(2);
    cl_barrier();                                                // own partials are in LDS; everyone is done with abuf[(it-1)&1]  (LDS-only fences: __syncthreads() put s_waitcnt vmcnt(0) -- the acknowledgement of the granule stores above -- in front of the barrier, and the polls behind it)
    // ---- polls first (the other members' partial tiles, in batches of <= PB members), bulk stores behind the first batch
    constexpr int NO = G - 1, PB = (NO > 4 || (RT > 1 && NO > 2)) ? (RT > 1 ? 2 : 4) : (NO > 0 ? NO : 1);
    const bool live = it > 0 && !dead;
    Pre cur;
    if (NO == 0 || !live) { if (it > 0) store_dz(tprev); else store_dz(t); wait_vm<NBULK>(); read_pre(cur); }
    f32x4 dh[RT];
#pragma unroll
    for (int k0 = 0; k0 < NO; k0 += PB) {
      u32x4 gr[PB][RT][2];
      auto issue = [&]() {
#pragma unroll
        for (int kk = 0; kk < PB; ++kk) {
          const int k = (k0 + kk) % (NO > 0 ? NO : 1), sm = k + (k >= member ? 1 : 0);   // the G-1 other members (straight-line: a tail batch repeats)
          const u64* src = pb + ((((size_t)(par * G + member) * G + sm) * 4 + wave) * RT) * 256 + lane_v * 2;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) poll16(gr[kk][rt][hf], src + rt * 256 + hf * 128);
        }
      };
      auto settle = [&]() {
#pragma unroll
        for (int kk = 0; kk < PB; ++kk)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) pin(gr[kk][rt][hf]);
      };
      if (!live) break;
      CL_STAMP(3);
      issue();
      if (k0 == 0) { store_dz(tprev); CL_STAMP(4); wait_vm<NBULK>(); read_pre(cur); }   // (step 0 stores zeros to the slot it rewrites one step later: handled above)
      else wait_vm<0>();
      settle();
      CL_STAMP(5);
      if (k0 == 0) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) dh[rt] = cur.dh[rt];
      }
      int spins = 0;
      while (true) {
        bool ok = true;
#pragma unroll
        for (int kk = 0; kk < PB; ++kk)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) ok = ok && gr[kk][rt][hf][1] == tag && gr[kk][rt][hf][3] == tag;
        if (__all(ok)) break;
        if (++spins > CL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(p.err, 2); break; }
        __builtin_amdgcn_s_sleep(1);
        issue(); wait_vm<0>(); settle();
      }
      if (dead) break;
#pragma unroll
      for (int kk = 0; kk < PB; ++kk) {
        if (k0 + kk >= NO) continue;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) { dh[rt][2 * hf] += __uint_as_float(gr[kk][rt][hf][0]); dh[rt][2 * hf + 1] += __uint_as_float(gr[kk][rt][hf][2]); }
      }
    }
    if (NO == 0 || !live) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) dh[rt] = cur.dh[rt];
    }
    CL_STAMP(6);
    if (it == 0 && d.dh2) {                                      // model.lua:667,681: d h of the decoder's initial state joins the first processed step
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = min(row0 + RH * rt + cv, B - 1);
        const f32x4 v = *reinterpret_cast<const f32x4*>(d.dh2 + (size_t)row * d.dh2_row + u0);
        dh[rt] += v;
      }
    }
    if (live && !dead) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) dh[rt] += own[(wave * RT + rt) * 64 + lane];
    }
    // ---- EpGatesBwd for this lane's cells: d z(t) as bf16 to LDS (next step's B operand); the HBM copies go out one step late
    unsigned char* an = abuf + (size_t)par * R * AP;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const bool ok = cok && row0 + RH * rt + c16 < B;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ig = cur.g[rt][i][0], fg = cur.g[rt][i][1], og = cur.g[rt][i][2], gg = cur.g[rt][i][3];
        const float tc = tanhf_(cur.cc[rt][i]);
        const float dc = dh[rt][i] * og * (1.f - tc * tc) + dcr[rt][i];
        const float d_o = dh[rt][i] * tc;
        const float z[4] = {dc * gg * ig * (1.f - ig), dc * cur.cp[rt][i] * fg * (1.f - fg), d_o * og * (1.f - og), dc * ig * (1.f - gg * gg)};
        dcr[rt][i] = dc * fg;
#pragma unroll
        for (int g = 0; g < 4; ++g) { zprev[rt][g][i] = z[g]; if (ok) dbs[g][i] += z[g]; }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 zz = {bf16_bits(zprev[rt][g][0]) | (bf16_bits(zprev[rt][g][1]) << 16), bf16_bits(zprev[rt][g][2]) | (bf16_bits(zprev[rt][g][3]) << 16)};
        *reinterpret_cast<u32x2*>(an + (size_t)(16 * rt + c16) * AP + (g * 64 + ul) * 2) = zz;
      }
    }
    CL_STAMP(7);
    cl_barrier();                                                // d z(t) complete in LDS
  }
#ifdef DC_DEBUG_STAMPS
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int k = 0; k < 8; ++k) p.err[16 + 2048 + 1020 + k] = (int)(cst_[k] >> 4);
#endif
  store_dz(d.forward_dir ? T - it1 : it1 - 1);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) { const int row = row0 + RH * rt + c16; if (cok && row < B) *reinterpret_cast<f32x4*>(d.dc + (size_t)row * He + u0) = dcr[rt]; }
  // bias gradients: both Linear layers see the same d z (LSTM.lua:79-88); sum over this lane's steps, then over the 16 rows of the tile
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = dbs[g][i];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (c16 == 0) { atomicAdd(d.dbi + g * He + u0 + i, v); atomicAdd(d.dbh + g * He + u0 + i, v); }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Every workgroup of a group must be resident at once (the polls wait for each other): a launch carries at most cus / G groups;
// more groups run as further launches (the groups are independent).  Two row tiles per group (RT = 2) halve the group count where the
// registers allow it (He <= 256).
bool enc_cluster_plan(int B, int He, int T, int cus, int& G, int& RT, int& groups) {
  if (He != 64 && He != 128 && He != 256 && He != 512) return false;
  if (T + 2 >= 4096 || B < 1 || cus < 8 * (He / 64)) return false;
  G = He / 64;
  RT = 1; groups = (B + 15) / 16;
  if (2 * groups * G > cus && G <= 4) { RT = 2; groups = (B + 31) / 32; }
  { static const char* e = getenv("AOCR_ENC_RT2"); if (e && e[0] == '1' && G <= 4 && B > 16) { RT = 2; groups = (B + 31) / 32; } }      // A/B: two row tiles per group although one fits the chip
  return true;
}
size_t enc_cluster_xbuf_bytes(int B, int He) {                   // forward exchange buffer for the largest plan (RT = 1 granularity covers RT = 2; 8-row groups: twice the slots)
  const int G = He / 64, groups = (B + 7) / 8;
  return (size_t)2 * groups * 2 * G * 32 * 32 * sizeof(u64);     // [gid][parity][member][<= 32 rows][32 granules]
}
size_t enc_cluster_pbuf_bytes(int B, int He) {
  const int G = He / 64, groups = (B + 7) / 8;                   // (8-row groups: twice the slots)
  return (size_t)2 * groups * 2 * G * G * 4 * 2 * 256 * sizeof(u64);
}

template <int G, int RT> static void launch_fwd(hipStream_t s, const EncClFwdArgs& a, int grid) {
  const size_t lds = (size_t)2 * 16 * RT * (64 * G * 2 + 16) + (size_t)4 * RT * 4 * 1024;
  if (a.d[0].ctx) hipLaunchKernelGGL((enc_cl_fwd_kernel<G, RT, true>), dim3(grid), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((enc_cl_fwd_kernel<G, RT, false>), dim3(grid), dim3(256), lds, s, a);
}
template <int G, int RT> static void launch_bwd(hipStream_t s, const EncClBwdArgs& a, int grid) {
  const size_t lds = (size_t)2 * 16 * RT * (256 * 2 + 16) + (size_t)4 * RT * 64 * 16 + (size_t)4 * RT * 7 * 1024;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)enc_cl_bwd_kernel<G, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((enc_cl_bwd_kernel<G, RT>), dim3(grid), dim3(256), lds, s, a);
}
static int cluster_cus() {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return cus;
}
void enc_cluster_forward(hipStream_t s, const EncClFwdArgs& a00, int G, int RT, int reserve_cus, int concurrent) {
  EncClFwdArgs a0 = a00; a0.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;    // testing aid: write-through granules even inside one XCD
  // Half tiles (8 batch rows per group) whenever twice the groups still fit the chip in one launch: the step is bound by the output stores of a CU
  // (store_outputs), and half the rows are half the bytes.  C3: 32 + 32 groups of 4 = all 256 CUs instead of 128; AOCR_ENC_RH16=1: 16-row groups.
  a0.rh = 16;
  { static const char* e = getenv("AOCR_ENC_RH16");
    const int g8 = (a0.B + 7) / 8;
    // (layer wavefront, round 6: `concurrent` launches of this size share the chip -- C5 at 128 strips: two layers x 128 CUs on 16-row groups overlap, two x 256 on 8-row groups would queue)
    if (!(e && e[0] == '1') && RT == 1 && a0.B > 8 && concurrent * 2 * g8 * G <= cluster_cus() - reserve_cus) { a0.rh = 8; a0.groups = g8; a0.gslot = a00.gslot * 2; } }
  const int per_pass = std::max(8, (cluster_cus() - reserve_cus) / (8 * G) * 8);          // groups (gids) one launch can keep resident; reserve_cus: compute units left to a co-resident collective (model.h: comm_reserved_cus)
  for (int g0 = 0; g0 < 2 * a0.groups; g0 += per_pass) {
    EncClFwdArgs a = a0; a.gid0 = g0; a.ngid = std::min(per_pass, 2 * a0.groups - g0);
    const int grid = 8 * G * ((a.ngid + 7) / 8);
#define AOCR_CL(GG) do { if (RT == 1) launch_fwd<GG, 1>(s, a, grid); else launch_fwd<GG, 2>(s, a, grid); } while (0)
    if (G == 1) AOCR_CL(1); else if (G == 2) AOCR_CL(2); else if (G == 4) AOCR_CL(4); else launch_fwd<8, 1>(s, a, grid);
#undef AOCR_CL
  }
}
void enc_cluster_backward(hipStream_t s, const EncClBwdArgs& a00, int G, int RT, int reserve_cus, int concurrent) {
  EncClBwdArgs a0 = a00; a0.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
  // Half tiles as in the forward launch, but only while they leave half the chip free: the hoisted weight-gradient GEMMs run beside this kernel on the
  // side streams (C3: 128 + 128 CUs; 8-row groups there would take all 256).  AOCR_ENC_BWD_RH=8 / 16 forces one.
  a0.rh = 16;
  { static const char* e = getenv("AOCR_ENC_BWD_RH");
    const int g8 = (a0.B + 7) / 8, cus = cluster_cus() - reserve_cus;
    const bool fits = RT == 1 && a0.B > 8 && concurrent * 2 * g8 * G <= cus;
    if (fits && ((e && e[0] == '8') || (!(e && e[0] == '1') && 2 * g8 * G <= cus / 2))) { a0.rh = 8; a0.groups = g8; a0.gslot = a00.gslot * 2; } }
  const int per_pass = std::max(8, (cluster_cus() - reserve_cus) / (8 * G) * 8);
  for (int g0 = 0; g0 < 2 * a0.groups; g0 += per_pass) {
    EncClBwdArgs a = a0; a.gid0 = g0; a.ngid = std::min(per_pass, 2 * a0.groups - g0);
    const int grid = 8 * G * ((a.ngid + 7) / 8);
#define AOCR_CL(GG) do { if (RT == 1) launch_bwd<GG, 1>(s, a, grid); else launch_bwd<GG, 2>(s, a, grid); } while (0)
    if (G == 1) AOCR_CL(1); else if (G == 2) AOCR_CL(2); else if (G == 4) AOCR_CL(4); else launch_bwd<8, 1>(s, a, grid);
#undef AOCR_CL
  }
}

}  // namespace aocr
