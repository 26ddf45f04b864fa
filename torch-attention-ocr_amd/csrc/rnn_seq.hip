// rnn_seq.hip -- whole-sequence ("persistent") kernels for the BiLSTM encoder recurrence (model.lua:291-316 forward,
// :662-690 backward; LSTM.lua:79-105 cell), bf16 operands / fp32 accumulate.
//
// The per-step kernels pay a launch + a cold weight fetch on every one of the T dependent steps (10-15 us each at
// B = 256, He = 256).  The recurrence is independent across batch rows, so here ONE workgroup owns 16 batch rows of one
// direction for all T steps: no inter-workgroup communication at all.  Per step it needs the whole recurrent weight
// (4He x He bf16 = 512 KB at He = 256), which does not fit on a CU, so it is re-streamed from L2 every step -- through
// LDS-DMA in full 128-byte lines (142 GB/s per CU measured, tools/ubench/wstream.hip; MFMA-fragment-shaped loads reach
// 38 GB/s) into wave-private rings, i.e. with no workgroup barrier on the weight path.  The stream never depends on
// h(t-1), so it runs ahead through the step boundary; only the 16 x He state operand is on the critical path.
//
// Wave w owns hidden units [32w, 32w+32) for all four gates (8 MFMA 16x16x32 column tiles): the gate non-linearities,
// the cell state (kept in registers for the whole sequence) and the recurrent gradient land in the lane that needs them.
// Every VMEM load in the loop is an LDS-DMA and every wait on them is a hand-counted s_waitcnt (static instruction
// counts: B % 16 == 0 so no lane is ever masked); the compiler never sees an ordinary VGPR-destination load there.
#include "ops.h"
#include <cstdlib>
#include <cstdio>

namespace aocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

#ifndef SEQ_RES_FWD
#define SEQ_RES_FWD 12               // weight units (of 32 per wave at He = 256) that never leave the registers, forward kernel
#endif
#ifndef SEQ_RES_BWD
#define SEQ_RES_BWD 4                // ... backward kernel (which also keeps 56 prefetched epilogue inputs in registers)
#endif
constexpr int SEQ_R = 4;                       // ring depth in units of 2 KiB (one 16-column tile x 64 k)
constexpr int SEQ_RING = SEQ_R * 2048;         // bytes per wave
constexpr int SEQ_ZXB = 8192;                  // per-wave staging of this step's pre-computed input part: 16 rows x 4 gates x 32 units fp32

__device__ __forceinline__ void seq_dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void seq_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); }
// Wait until streamed unit x (of ns) has landed.  Unit x is awaited two units ahead of its use: before the loop (pre: only the
// step's first R units have been issued) for x = 0, 1, otherwise in iteration x-2 right after unit x-2+R was issued.
// zx_pieces: VMEM pieces issued between the step's first R units and the units issued inside the loop (forward: 8).
template <int ZXP> __device__ __forceinline__ void seq_wait_units(int x, int ns, bool pre) {
  if (x < SEQ_R) { seq_wait_vm<63>(); return; }         // issued before the previous epilogue's >= 56 stores: more than 63 newer operations, so
                                                        // the 6-bit counter's maximum is a (stricter) valid wait
  int issued = pre ? SEQ_R : x - 2 + SEQ_R + 1;
  if (issued > ns) issued = ns;
  const int newer = 2 * (issued - 1 - x) + (x < SEQ_R ? ZXP : 0);
  switch (newer) {
    case 0: seq_wait_vm<0>(); break;
    case 2: seq_wait_vm<2>(); break;
    case 4: seq_wait_vm<4>(); break;
    case 6: seq_wait_vm<6>(); break;
    case 8: seq_wait_vm<8>(); break;
    case 10: seq_wait_vm<10>(); break;
    case 12: seq_wait_vm<12>(); break;
    default: seq_wait_vm<14>(); break;
  }
}
__device__ __forceinline__ void seq_wait_vm_after(int x, int ns, bool pre) { seq_wait_units<8>(x, ns, pre); }
__device__ __forceinline__ void seq_wait_vm_bwd(int x, int ns, bool pre) { seq_wait_units<0>(x, ns, pre); }
static_assert(SEQ_R == 4, "the wait helpers enumerate R - 1 = 3 newer units");

}  // namespace

// ---------------------------------------------------------------------------------------------
// forward: h(t), c(t) for all t of one direction; writes the state slots, the saved gates, the bf16 shadow of h and
// (top layer) the context slice -- exactly what the per-step EpGatesFwd epilogue writes.
// ---------------------------------------------------------------------------------------------
template <int NKS, bool CTX>
__global__ __launch_bounds__(512, 1) void enc_seq_fwd_kernel(EncSeqFwdArgs p) {
  constexpr int He = 32 * NKS, U = (NKS / 2) * 8, PITCH = He * 2 + 32;
  // The first RES units (k 0..63 of all 8 column tiles) never leave the registers (64 VGPRs); the other NS = U - RES are
  // streamed.  At He = 256 that cuts the per-step stream from 512 to 384 KB per workgroup.
  constexpr int RES = U > SEQ_RES_FWD ? SEQ_RES_FWD : (U > 8 ? 8 : 0), NS = U - RES;
  static_assert(NS % SEQ_R == 0, "streamed units must fill whole ring rounds");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];       // the ONLY LDS object
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const EncSeqDir& d = p.d[blockIdx.y];
  const int row0 = blockIdx.x * 16, B = p.B, T = p.T;
  unsigned char* const ring = lds + wave * SEQ_RING;
  unsigned char* const zxb = lds + 8 * SEQ_RING + wave * SEQ_ZXB;
  unsigned char* const hbuf = lds + 8 * SEQ_RING + 8 * SEQ_ZXB;                // [2][16][PITCH]
  const bool active = wave < NKS;                       // wave-uniform; idle waves only take part in the barriers

  // zero initial state h(-1) in both LDS buffers (the second one is overwritten by step 0)
  for (int i = tid; i < 2 * 16 * PITCH / 4; i += 512) reinterpret_cast<unsigned*>(hbuf)[i] = 0u;
  __syncthreads();

  if (active) {
    // ---- DMA source addressing.  W unit (kk, n): tile n = (gate n>>1, half n&1) rows, k in [64kk, 64kk+64) -> two 1-KiB
    // pieces of 8 rows x 128 B; lane L of piece pc covers row cc = 8pc + (L>>3), LDS position L&7, which holds the
    // 16-byte k-chunk (L&7) ^ ((cc>>1)&7) of that row (bank swizzle on the source side).
    // All global addresses are (wave-uniform 64-bit base) + (32-bit lane offset): one VGPR per stream instead of a
    // 64-bit pointer per instruction (the 64 weight pieces alone would otherwise pin 128 VGPRs across the step loop).
    unsigned wlane[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int cc = 8 * pc + (lane >> 3), chunk = (lane & 7) ^ ((cc >> 1) & 7);
      wlane[pc] = (unsigned)(((wave * 32 + cc) * He + 8 * chunk) * 2);
    }
    const char* wp0 = reinterpret_cast<const char*>(d.w) + wlane[0];
    const char* wp1 = reinterpret_cast<const char*>(d.w) + wlane[1];
    const int abl = p.abl;                              // timing-only ablations (AOCR_SEQ_ABL): 1 no weight DMA, 2 no stores, 4 no zx DMA
    auto issue_w = [&](int j) {                         // streamed unit j (compile-time after unrolling) = unit RES + j
      if (abl & 1) return;
      const int u = RES + j, kk = u >> 3, n = u & 7;
      const size_t off = (((size_t)(n >> 1) * He + (n & 1) * 16) * He + 64 * kk) * 2;
      unsigned char* dst = ring + (j % SEQ_R) * 2048;
      seq_dma16(wp0 + off, dst);
      seq_dma16(wp1 + off, dst + 1024);
    };
    bf16x8 wres[RES > 0 ? RES : 1][2];                  // resident fragments, loaded once (fragment-shaped loads: start-up only)
#pragma unroll
    for (int u = 0; u < RES; ++u)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kk = u >> 3, n = u & 7;
        wres[u][e] = *reinterpret_cast<const bf16x8*>(d.w + ((size_t)(n >> 1) * He + wave * 32 + (n & 1) * 16 + c16) * He + 64 * kk + 32 * e + 8 * q);
      }
    // zx piece pz (0..7): pairs (row r, gate g) = 8pz + (L>>3) -> r = pair>>2, g = pair&3; 128 B = this wave's 32 units;
    // LDS position (L&7) holds the 16-byte chunk (L&7) ^ (4 * ((r>>2)&1)): rows 4..7 / 12..15 sit half a bank row away.
    // pair = 8pz + (L>>3): r = 2pz + (L>>5), g = (L>>3)&3 -> one lane offset + a uniform 2-row stride per piece.
    const int zr = lane >> 5, zg = (lane >> 3) & 3;
    unsigned zlane[2];                                  // even / odd pz differ in ((r>>2)&1) only through 2pz: r>>2 = (2pz+zr)>>2
#pragma unroll
    for (int par = 0; par < 2; ++par) {                 // par = (pz>>1)&1 = (r>>2)&1 for r = 2pz + zr (zr < 2)
      const int chunk = (lane & 7) ^ (4 * par);
      zlane[par] = (unsigned)((((row0 + zr) * 4 * He) + zg * He + wave * 32 + 4 * chunk) * 4);
    }
    auto issue_zx = [&](int t) {
      if (abl & 4) return;
      const char* z = reinterpret_cast<const char*>(d.zx + (size_t)t * B * 4 * He);
#pragma unroll
      for (int pz = 0; pz < 8; ++pz) seq_dma16(z + (size_t)(2 * pz) * 4 * He * 4 + zlane[(pz >> 1) & 1], zxb + pz * 1024);
    };

    // fragment read offsets
    const int swz = (c16 >> 1) & 7;
    unsigned boff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) boff[e] = c16 * 128 + (((4 * e + q) ^ swz) << 4);
    const unsigned aoff = c16 * PITCH + q * 16;

    // lane offsets (elements) of cell (half 0, row 4q) in the state / context / gate tensors
    const unsigned lo_h = (unsigned)((row0 + 4 * q) * He + wave * 32 + c16);
    const unsigned ctx_rs = (unsigned)(T * p.Hd);
    const unsigned lo_c = (unsigned)(row0 + 4 * q) * ctx_rs + wave * 32 + c16;
    const unsigned lo_g = (unsigned)((row0 + 4 * q) * 4 * He + wave * 32 + c16);
    float cst[2][4];                                    // cell state of this lane's 8 (row, unit) cells
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) cst[s][i] = 0.f;

    // prologue: zx of step 0, then the first R weight units
    issue_zx(d.reverse ? T - 1 : 0);
#pragma unroll
    for (int j = 0; j < SEQ_R; ++j) issue_w(j);
    seq_wait_vm<0>();                                   // start-up only: the in-loop counts assume a full step of history

#define SEQ_STAMP(k) do { if (p.dbg && it == 10 && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0) p.dbg[k] = __builtin_readcyclecounter(); } while (0)
    for (int it = 0; it < T; ++it) {
      SEQ_STAMP(0);
      // The weight-piece addresses are step-invariant; left alone hipcc hoists all 64 of them out of the loop (128 VGPRs,
      // spills, and a vmcnt(0) per spill reload).  Laundering the two base pointers makes them per-step values.
      asm volatile("" : "+v"(wp0), "+v"(wp1));
      const int t = d.reverse ? T - 1 - it : it;
      const unsigned char* hcur = hbuf + (it & 1) * 16 * PITCH;
      unsigned char* hnxt = hbuf + ((it + 1) & 1) * 16 * PITCH;
      bf16x8 a[NKS];
#pragma unroll
      for (int s = 0; s < NKS; ++s) a[s] = *reinterpret_cast<const bf16x8*>(hcur + aoff + 64 * s);
      f32x4 acc[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < RES; ++u) {                   // register-resident units: no memory traffic at all
        const int kk = u >> 3, n = u & 7;
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * kk], wres[u][0], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * kk + 1], wres[u][1], acc[n], 0, 0, 0);
      }
      // Streamed units, fragment reads one unit ahead.  Streamed unit j sits in ring slot j % R.  The first R units of a
      // step were issued at the end of the previous epilogue (after its stores and zx pieces), unit j+R is issued in
      // iteration j: when unit x is awaited, min(NS, x+R) units have been issued, so 2 (min(NS, x+R) - 1 - x) pieces are
      // newer than it -- exact counts, no store in between.
      SEQ_STAMP(1);
      // fragments are read TWO units ahead (three register sets): one ds_read latency no longer fits between two units'
      // MFMAs.  Unit x is awaited in iteration x-2, right after unit x-2+R was issued.
      bf16x8 bs[3][2];
      seq_wait_vm_after(0, NS, true);
      SEQ_STAMP(2);
      bs[0][0] = *reinterpret_cast<const bf16x8*>(ring + boff[0]);
      bs[0][1] = *reinterpret_cast<const bf16x8*>(ring + boff[1]);
      if (NS > 1) {
        seq_wait_vm_after(1, NS, true);
        bs[1][0] = *reinterpret_cast<const bf16x8*>(ring + 2048 + boff[0]);
        bs[1][1] = *reinterpret_cast<const bf16x8*>(ring + 2048 + boff[1]);
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        if (j + 1 < NS) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // unit j's fragments are in registers (unit j+1's may be in flight)
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j + SEQ_R < NS) issue_w(j + SEQ_R);         // ... so its slot may be refilled
        if (j + 2 < NS) {
          seq_wait_vm_after(j + 2, NS, false);
          const unsigned char* slot = ring + ((j + 2) % SEQ_R) * 2048;
          bs[(j + 2) % 3][0] = *reinterpret_cast<const bf16x8*>(slot + boff[0]);
          bs[(j + 2) % 3][1] = *reinterpret_cast<const bf16x8*>(slot + boff[1]);
        }
        const int u = RES + j, kk = u >> 3, n = u & 7;
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * kk], bs[j % 3][0], acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * kk + 1], bs[j % 3][1], acc[n], 0, 0, 0);
      }
      SEQ_STAMP(3);
      // Restart the weight stream at once: the first R units of the next step land under the gate math (always issued, so
      // the counts stay uniform; after the last step they are drained below).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < SEQ_R && j < NS; ++j) issue_w(j);
      // ---- epilogue: this step's zx pieces are older than all NS weight units of this step and the R just issued
      seq_wait_vm<2 * (NS > SEQ_R ? NS - SEQ_R : 0) + 2 * (NS < SEQ_R ? NS : SEQ_R)>();  // only the units issued inside the loop are newer than the zx pieces
      float zx[4][2][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int r = 4 * q + i, u32 = s * 16 + c16;
            const int pos = (u32 >> 2) ^ (4 * ((r >> 2) & 1));
            zx[g][s][i] = *reinterpret_cast<const float*>(zxb + (r * 4 + g) * 128 + pos * 16 + (u32 & 3) * 4);
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      SEQ_STAMP(4);
      const size_t so = (size_t)(t + 1) * B * He;       // state slot t+1 holds step t
      float* const cs_t = d.cs + so; float* const hs_t = d.hs + so; bf16_t* const hb_t = d.hsb + so;
      float* const ctx_t = CTX ? d.ctx + (size_t)t * p.Hd : nullptr;
      float* const g_t = d.gates + (size_t)t * B * 4 * He;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          const float ig = sigmoidf_(acc[0 + s][i] + zx[0][s][i]), fg = sigmoidf_(acc[2 + s][i] + zx[1][s][i]);
          const float og = sigmoidf_(acc[4 + s][i] + zx[2][s][i]), gg = tanhf_(acc[6 + s][i] + zx[3][s][i]);
          const float cn = fg * cst[s][i] + ig * gg;
          const float hh = og * tanhf_(cn);
          cst[s][i] = cn;
          const unsigned o = lo_h + i * He + s * 16;
          if (!(abl & 2)) {
            cs_t[o] = cn; hs_t[o] = hh; hb_t[o] = (bf16_t)hh;
            if (CTX) ctx_t[lo_c + (unsigned)i * ctx_rs + s * 16] = hh;
            float* gp = g_t + (lo_g + i * 4 * He + s * 16);
            gp[0] = ig; gp[He] = fg; gp[2 * He] = og; gp[3 * He] = gg;
          }
          *reinterpret_cast<bf16_t*>(hnxt + r * PITCH + (wave * 32 + s * 16 + c16) * 2) = (bf16_t)hh;
        }
      __builtin_amdgcn_sched_barrier(0);
      SEQ_STAMP(5);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SEQ_STAMP(6);
      __builtin_amdgcn_s_barrier();                     // h(t) complete in LDS; everyone is done reading h(t-1)
      SEQ_STAMP(7);
      if (it + 1 < T) issue_zx(d.reverse ? t - 1 : t + 1); else issue_zx(t);     // needed one epilogue from now: off the critical path
      SEQ_STAMP(8);
    }
#undef SEQ_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // trailing stream pieces must land before the LDS is released
  } else {
    for (int it = 0; it < T; ++it) __builtin_amdgcn_s_barrier();
  }
}

bool enc_seq_supported(int B, int He, int blocks_limit) {
  return B % 16 == 0 && (He == 64 || He == 128 || He == 256) && (B / 16) * 2 <= blocks_limit;
}

void enc_seq_forward(hipStream_t s, const EncSeqFwdArgs& a0) {
  EncSeqFwdArgs a = a0;
  const char* e = getenv("AOCR_SEQ_ABL"); a.abl = e ? atoi(e) : 0;
  static unsigned long long* dbg = nullptr;
  const bool stamp = getenv("AOCR_SEQ_STAMP") != nullptr;
  if (stamp && !dbg) { (void)hipMalloc(&dbg, 16 * sizeof(unsigned long long)); }
  a.dbg = stamp ? dbg : nullptr;
  const int He = a.He;
  const size_t lds = 8 * SEQ_RING + 8 * SEQ_ZXB + 2 * 16 * (He * 2 + 32);
  dim3 grid(a.B / 16, 2), block(512);
  const bool ctx = a.d[0].ctx != nullptr;
#define AOCR_SEQ_FWD(NKS)                                                                                          \
  do {                                                                                                             \
    if (ctx) {                                                                                                     \
      (void)hipFuncSetAttribute((const void*)enc_seq_fwd_kernel<NKS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((enc_seq_fwd_kernel<NKS, true>), grid, block, lds, s, a);                                 \
    } else {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)enc_seq_fwd_kernel<NKS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      hipLaunchKernelGGL((enc_seq_fwd_kernel<NKS, false>), grid, block, lds, s, a);                                \
    }                                                                                                              \
  } while (0)
  if (He == 256) AOCR_SEQ_FWD(8); else if (He == 128) AOCR_SEQ_FWD(4); else AOCR_SEQ_FWD(2);
#undef AOCR_SEQ_FWD
  if (stamp) {                                          // debugging aid: cycle stamps of wave 0, workgroup (0,0), step 10
    unsigned long long h[16]; (void)hipStreamSynchronize(s); (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[aocr] enc_seq_fwd stamps (cycles since step top): A+resident %llu | unit0 landed %llu | loop done %llu | zx read %llu | math+stores %llu | ring issue %llu | barrier %llu | zx issue %llu\n",
            h[1] - h[0], h[2] - h[0], h[3] - h[0], h[4] - h[0], h[5] - h[0], h[6] - h[0], h[7] - h[0], h[8] - h[0]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (BPTT) of one layer: for every step, d z(t) from the saved gates / cell states and
// d h(t) = dh1(t) [+ dh2 at the first processed step] + d z(t') . Whh of the step t' processed just before -- exactly the
// per-step EpGatesBwd epilogue plus the recurrent GEMM in front of it.  The running d c stays in registers; d z(t) goes
// to HBM (fp32 + bf16, operands of the hoisted weight / input gradients) and, as bf16, into LDS as the A operand of the
// next step.  B operand: W^T ([He][4He] bf16, K = gate columns), streamed exactly like the forward weights.
// The epilogue's inputs (gates, c, c_prev, dh1: 56 floats per lane) are ordinary loads issued one step ahead, at the end
// of the previous epilogue; the compiler's own wait at their first use comes after the whole K loop, when every weight
// piece issued before them has long landed.
// ---------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(512, 1) void enc_seq_bwd_kernel(EncSeqBwdArgs p) {
  constexpr int He = 32 * NKS, KG = 4 * He, U = (KG / 64) * 2, APITCH = KG * 2 + 32;
  constexpr int RES = U > 8 ? SEQ_RES_BWD : 0, NS = U - RES;
  static_assert(NS % SEQ_R == 0, "streamed units must fill whole ring rounds");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const EncSeqBwdDir& d = p.d[blockIdx.y];
  const int row0 = blockIdx.x * 16, B = p.B, T = p.T;
  unsigned char* const ring = lds + wave * SEQ_RING;
  unsigned char* const abuf = lds + 8 * SEQ_RING;                              // [2][16][APITCH] bf16 d z of the previous step
  const bool active = wave < NKS;

  for (int i = tid; i < 2 * 16 * APITCH / 4; i += 512) reinterpret_cast<unsigned*>(abuf)[i] = 0u;   // "d z before the first step" = 0
  __syncthreads();

  if (active) {
    unsigned wlane[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int cc = 8 * pc + (lane >> 3), chunk = (lane & 7) ^ ((cc >> 1) & 7);
      wlane[pc] = (unsigned)(((wave * 32 + cc) * KG + 8 * chunk) * 2);
    }
    const char* wp0 = reinterpret_cast<const char*>(d.wt) + wlane[0];
    const char* wp1 = reinterpret_cast<const char*>(d.wt) + wlane[1];
    auto issue_w = [&](int j) {                         // streamed unit j = unit RES + j = (kk = u >> 1, tile n = u & 1)
      const int u = RES + j, kk = u >> 1, n = u & 1;
      const size_t off = ((size_t)n * 16 * KG + 64 * kk) * 2;
      unsigned char* dst = ring + (j % SEQ_R) * 2048;
      seq_dma16(wp0 + off, dst);
      seq_dma16(wp1 + off, dst + 1024);
    };
    bf16x8 wres[RES > 0 ? RES : 1][2];
#pragma unroll
    for (int u = 0; u < RES; ++u)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kk = u >> 1, n = u & 1;
        wres[u][e] = *reinterpret_cast<const bf16x8*>(d.wt + (size_t)(wave * 32 + n * 16 + c16) * KG + 64 * kk + 32 * e + 8 * q);
      }
    const int swz = (c16 >> 1) & 7;
    unsigned boff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) boff[e] = c16 * 128 + (((4 * e + q) ^ swz) << 4);
    const unsigned aoff = c16 * APITCH + q * 16;

    const unsigned lo_h = (unsigned)((row0 + 4 * q) * He + wave * 32 + c16);
    const unsigned lo_g = (unsigned)((row0 + 4 * q) * KG + wave * 32 + c16);
    const unsigned lo_d = (unsigned)(row0 + 4 * q) * (unsigned)d.dh1_row + wave * 32 + c16;
    const unsigned dh1_rs = (unsigned)d.dh1_row;

    float dcr[2][4];                                    // running d c of this lane's 8 cells
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) dcr[s][i] = d.dc[lo_h + i * He + s * 16];

    float pg[4][2][4], pcc[2][4], pcp[2][4], pdh[2][4];    // the next step's epilogue inputs
    auto prefetch = [&](int it) {
      const int t = d.forward_dir ? T - 1 - it : it, prev = d.forward_dir ? t : t + 2;
      const float* g_t = d.gates + (size_t)t * B * KG;
      const float* c_t = d.cs + (size_t)(t + 1) * B * He;
      const float* cp_t = d.cs + (size_t)prev * B * He;
      const float* dh_t = d.dh1 + (size_t)t * d.dh1_t;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int g = 0; g < 4; ++g) pg[g][s][i] = g_t[lo_g + i * KG + s * 16 + g * He];
          pcc[s][i] = c_t[lo_h + i * He + s * 16];
          pcp[s][i] = cp_t[lo_h + i * He + s * 16];
          pdh[s][i] = dh_t[lo_d + (unsigned)i * dh1_rs + s * 16];
        }
    };
    prefetch(0);
    if (d.dh2) {                                        // model.lua:667,681: d h of the decoder's initial state joins the first processed step
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) pdh[s][i] += d.dh2[(size_t)(row0 + 4 * q + i) * d.dh2_row + wave * 32 + s * 16 + c16];
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < SEQ_R && j < NS; ++j) issue_w(j);
    seq_wait_vm<0>();

    for (int it = 0; it < T; ++it) {
      asm volatile("" : "+v"(wp0), "+v"(wp1));          // see enc_seq_fwd_kernel: keeps the 64 piece addresses out of registers
      const int t = d.forward_dir ? T - 1 - it : it;
      const unsigned char* acur = abuf + (it & 1) * 16 * APITCH;
      unsigned char* anxt = abuf + ((it + 1) & 1) * 16 * APITCH;
      f32x4 acc[2][2];
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int e = 0; e < 2; ++e) acc[n][e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < RES; ++u) {
        const int kk = u >> 1, n = u & 1;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(acur + aoff + 128 * kk);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(acur + aoff + 128 * kk + 64);
        acc[n][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, wres[u][0], acc[n][0], 0, 0, 0);
        acc[n][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, wres[u][1], acc[n][1], 0, 0, 0);
      }
      bf16x8 bs[3][2];                                  // weight fragments two units ahead (see enc_seq_fwd_kernel)
      seq_wait_vm_bwd(0, NS, true);
      bs[0][0] = *reinterpret_cast<const bf16x8*>(ring + boff[0]);
      bs[0][1] = *reinterpret_cast<const bf16x8*>(ring + boff[1]);
      if (NS > 1) {
        seq_wait_vm_bwd(1, NS, true);
        bs[1][0] = *reinterpret_cast<const bf16x8*>(ring + 2048 + boff[0]);
        bs[1][1] = *reinterpret_cast<const bf16x8*>(ring + 2048 + boff[1]);
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) {
        const int u = RES + j, kk = u >> 1, n = u & 1;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(acur + aoff + 128 * kk);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(acur + aoff + 128 * kk + 64);
        if (j + 1 < NS) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");   // unit j's weight fragments are back (newer: unit j+1's pair, this A pair)
        else asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        if (j + SEQ_R < NS) issue_w(j + SEQ_R);
        if (j + 2 < NS) {
          seq_wait_vm_bwd(j + 2, NS, false);
          const unsigned char* slot = ring + ((j + 2) % SEQ_R) * 2048;
          bs[(j + 2) % 3][0] = *reinterpret_cast<const bf16x8*>(slot + boff[0]);
          bs[(j + 2) % 3][1] = *reinterpret_cast<const bf16x8*>(slot + boff[1]);
        }
        acc[n][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bs[j % 3][0], acc[n][0], 0, 0, 0);
        acc[n][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bs[j % 3][1], acc[n][1], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < SEQ_R && j < NS; ++j) issue_w(j);          // the next step's first R units land under the epilogue
      // ---- epilogue (EpGatesBwd): the inputs were loaded one step ahead
      float* const dz_t = d.dz + (size_t)t * B * KG;
      bf16_t* const dzb_t = d.dzb + (size_t)t * B * KG;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          const float dh = acc[s][0][i] + acc[s][1][i] + pdh[s][i];
          const float ig = pg[0][s][i], fg = pg[1][s][i], og = pg[2][s][i], gg = pg[3][s][i];
          const float tc = tanhf_(pcc[s][i]);
          const float dc = dh * og * (1.f - tc * tc) + dcr[s][i];
          const float d_o = dh * tc;
          const float di = dc * gg, dg = dc * ig, df = dc * pcp[s][i];
          const float z[4] = {di * ig * (1.f - ig), df * fg * (1.f - fg), d_o * og * (1.f - og), dg * (1.f - gg * gg)};
          dcr[s][i] = dc * fg;
          const unsigned o = lo_g + i * KG + s * 16;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            dz_t[o + g * He] = z[g];
            dzb_t[o + g * He] = (bf16_t)z[g];
            *reinterpret_cast<bf16_t*>(anxt + r * APITCH + (g * He + wave * 32 + s * 16 + c16) * 2) = (bf16_t)z[g];
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      prefetch(it + 1 < T ? it + 1 : it);               // always issued (uniform code); the last one is simply unused
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // d z(t) complete in LDS; everyone is done reading the previous one
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) d.dc[lo_h + i * He + s * 16] = dcr[s][i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int it = 0; it < T; ++it) __builtin_amdgcn_s_barrier();
  }
}

void enc_seq_backward(hipStream_t s, const EncSeqBwdArgs& a) {
  const int He = a.He;
  const size_t lds = 8 * SEQ_RING + 2 * 16 * (4 * He * 2 + 32);
  dim3 grid(a.B / 16, 2), block(512);
#define AOCR_SEQ_BWD(NKS)                                                                                          \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)enc_seq_bwd_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((enc_seq_bwd_kernel<NKS>), grid, block, lds, s, a);                                         \
  } while (0)
  if (He == 256) AOCR_SEQ_BWD(8); else if (He == 128) AOCR_SEQ_BWD(4); else AOCR_SEQ_BWD(2);
#undef AOCR_SEQ_BWD
}

}  // namespace aocr
