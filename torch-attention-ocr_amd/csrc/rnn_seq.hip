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

namespace aocr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SEQ_R = 4;                       // ring depth in units of 2 KiB (one 16-column tile x 64 k)
constexpr int SEQ_RING = SEQ_R * 2048;         // bytes per wave
constexpr int SEQ_ZXB = 8192;                  // per-wave staging of this step's pre-computed input part: 16 rows x 4 gates x 32 units fp32

__device__ __forceinline__ void seq_dma16(const void* g, unsigned char* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void seq_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory"); }

}  // namespace

// ---------------------------------------------------------------------------------------------
// forward: h(t), c(t) for all t of one direction; writes the state slots, the saved gates, the bf16 shadow of h and
// (top layer) the context slice -- exactly what the per-step EpGatesFwd epilogue writes.
// ---------------------------------------------------------------------------------------------
template <int NKS, bool CTX>
__global__ __launch_bounds__(512, 1) void enc_seq_fwd_kernel(EncSeqFwdArgs p) {
  constexpr int He = 32 * NKS, U = (NKS / 2) * 8, PITCH = He * 2 + 32;
  constexpr int STORES = CTX ? 64 : 56;                 // VMEM stores per lane per step
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
    auto issue_w = [&](int u) {                         // u compile-time after unrolling
      const int kk = u >> 3, n = u & 7;
      const size_t off = (((size_t)(n >> 1) * He + (n & 1) * 16) * He + 64 * kk) * 2;
      unsigned char* dst = ring + (u % SEQ_R) * 2048;
      seq_dma16(wp0 + off, dst);
      seq_dma16(wp1 + off, dst + 1024);
    };
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
    for (int u = 0; u < SEQ_R; ++u) issue_w(u);
    seq_wait_vm<0>();                                   // start-up only: the in-loop counts assume a full step of history

    for (int it = 0; it < T; ++it) {
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
      for (int u = 0; u < U; ++u) {
        // unit u was issued R units ago; after it: the R-1 later units, and for the first R units of a step also the
        // previous epilogue's zx pieces and stores (capped at the 6-bit counter: a stricter wait, never a looser one)
        if (u < SEQ_R) seq_wait_vm<2 * (SEQ_R - 1) + 8 + STORES>(); else seq_wait_vm<2 * (SEQ_R - 1)>();
        const unsigned char* slot = ring + (u % SEQ_R) * 2048;
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(slot + boff[0]);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(slot + boff[1]);
        const int kk = u >> 3, n = u & 7;
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * kk], b0, acc[n], 0, 0, 0);
        acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2 * kk + 1], b1, acc[n], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are back before it is refilled
        issue_w((u + SEQ_R) % U);                       // the stream wraps into the next step: the weights do not change
      }
      // ---- epilogue: this step's zx pieces were issued one epilogue ago, 2U weight pieces (+ stores) later
      seq_wait_vm<63>();
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
      if (it + 1 < T) issue_zx(d.reverse ? t - 1 : t + 1); else issue_zx(t);      // always 8 pieces: uniform counts
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
          cs_t[o] = cn; hs_t[o] = hh; hb_t[o] = (bf16_t)hh;
          if (CTX) ctx_t[lo_c + (unsigned)i * ctx_rs + s * 16] = hh;
          float* gp = g_t + (lo_g + i * 4 * He + s * 16);
          gp[0] = ig; gp[He] = fg; gp[2 * He] = og; gp[3 * He] = gg;
          *reinterpret_cast<bf16_t*>(hnxt + r * PITCH + (wave * 32 + s * 16 + c16) * 2) = (bf16_t)hh;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                     // h(t) complete in LDS; everyone is done reading h(t-1)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // trailing stream pieces must land before the LDS is released
  } else {
    for (int it = 0; it < T; ++it) __builtin_amdgcn_s_barrier();
  }
}

bool enc_seq_supported(int B, int He, int blocks_limit) {
  return B % 16 == 0 && (He == 64 || He == 128 || He == 256) && (B / 16) * 2 <= blocks_limit;
}

void enc_seq_forward(hipStream_t s, const EncSeqFwdArgs& a) {
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
}

}  // namespace aocr
