// dec_cluster.hip -- the teacher-forced decoder loop (model.lua:553-568 train, :604-627 gold pass; cell LSTM.lua:18-122, attention
// LSTM.lua:124-162) as ONE launch for all L steps: a GROUP of 32 compute units (one XCD under round-robin dispatch) owns 32 batch
// rows, and every recurrent weight the loop needs -- W1 = [W1_i2h(feed part) | W1_h2h], W2 = [W2_i2h | W2_h2h], W_c: 9 MB in bf16
// at Hd = 512 -- is RESIDENT IN THE REGISTERS of the group (288 VGPRs per lane: member m owns hidden units 16m .. 16m+15 of both
// layers and output columns 16m .. 16m+15 of W_c).  The launch chain this replaces runs 5 dependent kernels per step, each of
// which re-streams its weights through L2 and pays a launch + drain (~8 us each at C3: 0.94 ms for 24 steps).
// The same file holds the BPTT of the loop (dec_cl_bwd_kernel) and the greedy decode variant (dec_cl_fwd_kernel<true>).
//
// Per step, four all-gathers of a 32 x 512 bf16 operand inside the group.  The payload of an exchange is the real output tensor
// (h, [c ; h], out as bf16): every wave stores its piece, waits for the acknowledgement and raises its flag; a reader polls the 128
// flags of the group (512 bytes) and then loads the operand once (`gather`).  (The first version polled 8-byte {2 x bf16, tag}
// granules as rnn_cluster.hip does: 64 KB per poll round and CU, bound by the CU's 64 B/clk vector-memory path.)
//   out(t-1) -> [feed ; h1(t-1)] W1^T + zx1(t) -> gates -> c1, h1          (zx1 = embedding part + biases, hoisted over all L steps)
//   h1(t)    -> [h1(t) ; h2(t-1)] W2^T + b -> gates -> c2, h2
//   h2(t)    -> attention of row r on member r (q = W_a h2, which only the backward pass reads, is one GEMM over all L steps after the loop):
//               s = ctxA[r] . h2  (ctxA = ctx . W_a precomputed once: ctx . (W_a h) = (ctx W_a) . h), a = softmax(s), c = a . ctx[r]
//   c(t)     -> out = tanh(W_c [c ; h2])
// Operands live in LDS ([32 rows][512] bf16 x 3 buffers); the weights of a wave are MFMA A fragments (transposed products, as in
// rnn_cluster.hip), tile rows ordered [unit][gate] so that a lane holds the four gates of one (unit, batch row) cell.
// What the backward pass reads is written for all L steps: gates (interleaved per unit), cell states, a, out in fp32; h and [c ; h2]
// as bf16 only (every reader takes the bf16 copy).  The launch chain stays the fallback (fp32 mode, Hd != 512, other layer counts,
// no input feed, dropout > 0, beam > 1) and the parity reference (tests/test_step_gpu.py).
#include "ops.h"
#include "dec_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace aocr {

namespace {
#ifdef DC_DEBUG_STAMPS
#define DC_TL(cond, k) do { if (tl && (cond)) tl[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DC_GS(k) do { if (stamps) { const u64 t1_ = __builtin_readcyclecounter(); gs[k] += t1_ - t0; t0 = t1_; } } while (0)
#else
#define DC_TL(cond, k) do { } while (0)
#define DC_GS(k) do { } while (0)
#endif
// All-gather of one 32 x 512 bf16 operand into an LDS buffer.  The payload is the real output tensor (h, [c ; h] or out as bf16): every
// wave of every member stores its piece, waits for the write acknowledgement, then raises its flag (128 flags of 4 bytes per
// operand); a reader polls the flags -- 512 bytes per round instead of the operand -- and loads the 32 KB once.
//   payload():  this wave's piece (any number of stores);   deferred(): exactly NST one-instruction stores of older outputs, issued
//   behind the operand loads (the flag polls wait for everything older than them).
template <int NST, class PAY, class DEF>
__device__ __forceinline__ void gather(unsigned* flags, unsigned tag, const bf16_t* src, int stride_bytes, int row0, int B, unsigned char* dst, int tid, int wave,
                                       int member, bool local, int* err, int code, int* dead_flag, PAY&& payload, DEF&& deferred, u64 (&gs)[3], bool stamps, [[maybe_unused]] u64* tl) {
  [[maybe_unused]] u64 t0 = stamps ? __builtin_readcyclecounter() : 0;
  payload();
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the piece is in L2.  (The builtin, not asm: the compiler's own count of outstanding loads restarts here,
                                                //  so it inserts no waits of its own between the asm loads below.)
  if ((tid & 63) < NM) pst4(flags + (size_t)(tid & 63) * 128 + member * 4 + wave, tag, local);      // one copy per reader: polls of different members never meet in one line
  DC_TL((tid & 63) == 0, wave);
  DC_GS(0);
  const unsigned foff = (unsigned)(member * 128 + (tid & 63)) * 4;
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
  DC_GS(1);
  DC_TL((tid & 63) == 0, 4 + wave);
  u32x4 gr[8];                                  // 32 rows x 64 chunks of 16 bytes: thread -> chunk tid & 63 of rows (tid >> 6) + 4 j
#pragma unroll
  for (int j = 0; j < 8; ++j) ld16_sc1(gr[j], (unsigned)(min(row0 + (((tid >> 6) + 4 * j + member) & 31), B - 1) * stride_bytes + (tid & 63) * 16), src, local);   // (members start at different rows)
  deferred();                                   // behind the loads: nothing on the exchange path waits for their acknowledgements
  DC_TL((tid & 63) == 0, 8 + wave);
  lds_barrier();                                // every wave of this workgroup is past its reads of the previous contents of dst
  DC_TL(tid == 0, 12);
  wait_vm<NST>();
  DC_TL(tid == 0, 13);
#pragma unroll
  for (int j = 0; j < 8; ++j) dpin(gr[j]);
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(dst + (size_t)(((tid >> 6) + 4 * j + member) & 31) * PA + (tid & 63) * 16) = gr[j];
  lds_barrier();
  DC_TL(tid == 0, 14);
  DC_GS(2);
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
template <int CH, int NST, class DEF>
__device__ __forceinline__ void fetch_rows(const bf16_t* src, int stride_bytes, int row0, int B, unsigned char* dst, int pitch, int tid, int member, bool local, DEF&& deferred) {
  // 32 rows x CH KB into LDS (row pitch `pitch`), CH passes of 8 loads per thread, two passes in flight; deferred(): exactly NST stores
  // behind the first pass
  u32x4 g0[8], g1[8];
  auto issue = [&](u32x4 (&g)[8], int c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ld16_sc1(g[j], (unsigned)(min(row0 + (((tid >> 6) + 4 * j + member) & 31), B - 1) * stride_bytes + c * 1024 + (tid & 63) * 16), src, local);
  };
  auto land = [&](u32x4 (&g)[8], int c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) dpin(g[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(dst + (size_t)(((tid >> 6) + 4 * j + member) & 31) * pitch + c * 1024 + (tid & 63) * 16) = g[j];
  };
  issue(g0, 0);
  if constexpr (CH == 1) {
    deferred();
    lds_barrier();                              // every wave of this workgroup is past its reads of the previous contents of dst
    wait_vm<NST>(); land(g0, 0);
  } else {
    static_assert(CH == 4, "1 or 4 KB rows");
    issue(g1, 1);
    lds_barrier();
    wait_vm<8>(); land(g0, 0);
    issue(g0, 2); wait_vm<8>(); land(g1, 1);
    issue(g1, 3); wait_vm<8>(); land(g0, 2);
    deferred();                                 // behind the last pass: no load waits for these acknowledgements
    wait_vm<NST>(); land(g1, 3);
  }
  lds_barrier();
}
// payload(): this wave's piece; then the acknowledgement, the flag (one copy per reader) and the wait for all 128 flags of the group
template <class PAY>
__device__ __forceinline__ void raise_and_wait(unsigned* flags, unsigned tag, int tid, int wave, int member, bool local, int* err, int code, int* dead_flag, PAY&& payload) {
  payload();
  __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0): the piece is in L2 (and the compiler's own count restarts: see gather)
  if ((tid & 63) < NM) pst4(flags + (size_t)(tid & 63) * 128 + member * 4 + wave, tag, local);
  const unsigned foff = (unsigned)(member * 128 + (tid & 63)) * 4;
  unsigned f0, f1; int spins = 0;
#pragma nounroll
  while (true) {
    ld4_sc1(f0, foff, flags, local); ld4_sc1(f1, foff + 256, flags, local);
    wait_vm<0>();
    asm volatile("" : "+v"(f0), "+v"(f1));
    if (__all(f0 == tag && f1 == tag)) break;
    asm volatile("" : "+s"(spins));
    if (++spins > DC_SPIN_LIMIT) { if ((tid & 63) == 0) { atomicExch(err, code); *dead_flag = 1; } break; }
    __builtin_amdgcn_s_sleep(1);
  }
}

}  // namespace

// Debugging aids, compiled in with -DDC_DEBUG_STAMPS only (their global stores make the compiler insert vmcnt waits between the asm loads):
// cycles per phase of workgroup 0 and a real-time timeline of one step of group 0 (tools/dc_stamp.py).
#ifdef DC_DEBUG_STAMPS
#define DC_STAMP(k) do { if (p.stamps) { const u64 now_ = __builtin_readcyclecounter(); stamp[k] += now_ - tprev; tprev = now_; } } while (0)
#else
#define DC_STAMP(k) do { } while (0)
#endif

// DEC = false: teacher-forced loop (train step, gold pass).  DEC = true: greedy decode (model.lua:376-536 at beam 1): the token fed to step
// t+1 is the arg-max of step t -- the projector + LogSoftMax + selection of project_select_kernel run inside the loop: every member
// multiplies its 16 fp32 units of out(t) with its slice of W_o, the partial logits of row r meet on member r (one more exchange),
// which selects, keeps the running score, writes the label and publishes the token with its next out(t) flag.
// DROP: nn.Dropout(p > 0) of the training step (LSTM.lua:68-69: layer 2 reads Dropout(h1); :116-118: Dropout on the attention output).
// The masks are the counter-based function of epilogues.h::DropSpec the launch chain uses (same sites, same indices).  h1(t) is
// published twice -- as it is (the recurrence of layer 1 reads it at step t+1) and masked (hm_b: the operand of layer 2, gathered into
// the H1 buffer; H1 is re-fetched unmasked at the start of the next step) --, out(t) only masked (projector, input feed and the
// backward kernel all see Dropout(out); the backward kernel divides the mask out again for tanh').
template <bool DEC, bool DROP = false>
__global__ __launch_bounds__(256, 1) void dec_cl_fwd_kernel(DecClFwdArgs p) {
  static_assert(!(DEC && DROP), "evaluate(): no dropout in decode");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const F = lds;                        // feed (out(t-1)), later c(t)
  unsigned char* const H1 = lds + R * PA;              // h1(t-1), later h1(t)
  unsigned char* const H2 = lds + 2 * R * PA;          // h2(t-1), later h2(t)
  float* const red = reinterpret_cast<float*>(lds + 3 * R * PA);     // [4 waves][4 tiles][2][64 lanes][4]: K-split partial tiles (32 KB);
  float* const part = red;                                            // attention: [4 waves][512] partial context,
  float* const sc = red + 2048;                                       //            [256] scores
  float* const wos = reinterpret_cast<float*>(lds + 3 * R * PA + 32768);      // DEC: [40][16] this member's slice of W_o (fp32)
  float* const outs = wos + 640;                                              //      [32][16] out(t) of this member's units (fp32)
  int* const toks = reinterpret_cast<int*>(outs + 512);                       //      [32] the tokens fed to the current step
  float* const ztab = reinterpret_cast<float*>(toks + 32);                    //      [40 tokens][4 gates][16] this member's columns of the per-token gate-input table
  __shared__ int s_local, s_dead, s_allfin;
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
  float c1[2], c2[2], zxr[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int row = min(row0 + 16 * rt + c16, B - 1);
    c1[rt] = p.cs[0][(size_t)row * HD + unit]; c2[rt] = p.cs[1][(size_t)row * HD + unit];
    const size_t zrow = DEC ? (size_t)(p.tok0[(size_t)row * p.tok0_stride] - 1)                      // DEC: the per-token table row of the GO token
                            : (p.zx_tok ? (size_t)(min(max(p.zx_tok[(int64_t)row * p.zx_sb], 1), p.V) - 1) : (size_t)row);   // teacher-forced with the table: the token of step 0
#pragma unroll
    for (int i = 0; i < 4; ++i) zxr[rt][i] = p.zx1[zrow * 4 * HD + i * HD + unit];
  }
  int ntok[2] = {1, 1};                                               // teacher-forced with the table: the tokens of the NEXT step (loaded a step ahead: no dependent load on the step's path)
  if constexpr (!DEC) {
    if (p.zx_tok) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) ntok[rt] = p.zx_tok[(int64_t)min(1, L - 1) * p.zx_st + (int64_t)min(row0 + 16 * rt + c16, B - 1) * p.zx_sb];
    }
  }
  if constexpr (DEC) {
    for (int i = tid; i < 640; i += 256) { const int v = i >> 4, u = i & 15; wos[i] = v < p.V ? p.wo[(size_t)v * HD + 16 * member + u] : 0.f; }
    for (int i = tid; i < 2560; i += 256) {            // 10 KB: a step's table lookup is an LDS read behind the token, not a second L2 round trip
      const int v = i >> 6, g = (i >> 4) & 3, u = i & 15;
      ztab[i] = v < p.V ? p.zx1[(size_t)v * 4 * HD + g * HD + 16 * member + u] : 0.f;
    }
    if (tid < 32) toks[tid] = p.tok0[(size_t)min(row0 + tid, B - 1) * p.tok0_stride];
  }
  unsigned* const xg = reinterpret_cast<unsigned*>(p.xbuf) + (size_t)group * 5 * NM * 128;      // flags [kind: out, h1, h2, c, logits][reader][member][wave]
  int t_exit = -1;                                                  // DEC: the step at which every row of the group had finished (early exit)
  float score = 0.f; int prev_tok = 0, node = 0;                    // DEC, wave 0 of the row's owner: running log-probability, last token, trie node (-use_dictionary)
  load_rows(p.out_b, row0, B, F, tid); load_rows(p.hsb[0], row0, B, H1, tid); load_rows(p.hsb[1], row0, B, H2, tid);
  __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): nothing of the prologue is in flight inside the loop
  __syncthreads();
  u64 stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, gs[3] = {0, 0, 0}, tprev = __builtin_readcyclecounter();

  // the row of the group whose attention this workgroup computes
  const int arow = row0 + member; const bool rvalid = arow < B;
  const bf16_t* const ca = p.ctxa + (size_t)min(arow, B - 1) * T * HD;
  const bf16_t* const cx = p.ctxb + (size_t)min(arow, B - 1) * T * HD;
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
  u32x2 hpm[2] = {u32x2{0u, 0u}, u32x2{0u, 0u}};                    // DROP: Dropout(h1) of this lane's cells, packed like hp
  auto cell = [&](const f32x4 (&z)[2], float (&c)[2], f32x4 (&g)[2], u32x2 (&hp)[2], long long moff = -1) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float ig = sigmoidf_(z[rt][0]), fg = sigmoidf_(z[rt][1]), og = sigmoidf_(z[rt][2]), gg = tanhf_(z[rt][3]);
      const float cn = fg * c[rt] + ig * gg, hn = og * tanhf_(cn);
      c[rt] = cn; g[rt] = f32x4{ig, fg, og, gg};
      const unsigned hb = bfbits(hn);
      const unsigned h1v = __shfl(hb, lane + 16, 64), h2v = __shfl(hb, lane + 32, 64), h3v = __shfl(hb, lane + 48, 64);
      hp[rt] = u32x2{hb | (h1v << 16), h2v | (h3v << 16)};
      if constexpr (DROP) {
        if (moff >= 0) {                                              // mask index = step offset + row * Hd + unit (EpGatesFwd::drop)
          DropSpec d = p.drop_h; d.off = moff;
          const unsigned mb = bfbits(hn * d.mask((long long)(row0 + 16 * rt + c16) * HD + unit));
          const unsigned m1 = __shfl(mb, lane + 16, 64), m2 = __shfl(mb, lane + 32, 64), m3 = __shfl(mb, lane + 48, 64);
          hpm[rt] = u32x2{mb | (m1 << 16), m2 | (m3 << 16)};
        }
      }
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

    u64* const tl0 = (p.stamps && group == 0 && t == 10) ? reinterpret_cast<u64*>(p.err + 16 + 2048) + member * 64 : nullptr;   // timeline of one step (debugging aid)
    int ot = tid; asm volatile("" : "+v"(ot));                     // opaque copy of the thread id (see store_layer)
    const int olane = ot & 63, oc16 = ot & 15, oq = (ot >> 4) & 3, ounit = 16 * member + 4 * wave + oq;
    unsigned char* const otrash = reinterpret_cast<unsigned char*>(p.err + 16) + ot * 16;
    // =================== layer 1: z1 = [feed ; h1(t-1)] W1^T + zx1(t)
    if (t > 0) {
      gather<1>(xg + 0 * NM * 128, tagc - 1u, p.out_b + (size_t)t * slot, HD * 2, row0, B, F, ot, wave, member, local, p.err, 11, &s_dead,
                [&] {                                               // out(t-1) of this member's 16 units (waves 0, 1: one row tile each)
                  const int row = row0 + 16 * wave + oc16;
                  pst8(wave < 2 && row < B ? (void*)(p.out_b + (size_t)t * slot + (size_t)row * HD + 16 * member + 4 * oq) : (void*)otrash, ovb, local);
                },
                [&] { store_out(ot, t - 1); }, gs, p.stamps != nullptr, tl0);
      if (s_dead) break;
    }
    if constexpr (DROP) {                           // H1 holds Dropout(h1(t-1)) (layer 2's operand): layer 1's recurrence needs h1(t-1) itself
      if (t > 0) fetch_rows<1, 0>(p.hsb[0] + (size_t)t * slot, HD * 2, row0, B, H1, PA, ot, member, local, [] {});      // (all 128 flags of h1(t-1) were seen a step ago)
    }
    unsigned tk = 0;
    if constexpr (DEC) {                            // the tokens chosen at step t-1 (published before the owners' out flags): the load is
      if (t > 0 && wave == 0) {                     // issued here and consumed AFTER the layer-1 products, which do not depend on it
        unsigned* const tp = p.tokx + (size_t)group * 32 + (ot & 31);
        tk = local ? __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    DC_STAMP(0);
    f32x4 g1[2]; u32x2 hp1[2];
    {
      f32x4 z[2];
      product(w1a, w1b, F, H1, z);
      if constexpr (DEC) {
        if (t > 0) {
          if (wave == 0) {
            if (ot < 32) toks[ot] = (int)tk;
            // Every row of the group has emitted EOS (or PAD): from here on each step selects PAD at no cost (model.lua:448-449), so the
            // labels of the remaining steps are PAD and the scores final (all members see the same 32 tokens, so all leave together).
            const bool done = tk == 1u || tk == 3u || row0 + (ot & 31) >= B;
            const unsigned long long all = __ballot(done);
            if (ot == 0) s_allfin = all == ~0ull;
          }
          lds_barrier();
          if (s_allfin && !p.no_early) { t_exit = t; break; }       // (the PAD labels / final scores are written behind the loop)
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            const int zrow = min(max(toks[16 * rt + oc16], 1), p.V) - 1;        // (clamped: a corrupted token must not become a wild address)
#pragma unroll
            for (int i = 0; i < 4; ++i) zxr[rt][i] = ztab[zrow * 64 + i * 16 + 4 * wave + oq];
          }
        }
      }
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) z[rt][i] += zxr[rt][i];
      cell(z, c1, g1, hp1, DROP ? (long long)t * (long long)slot : -1);
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
    gather<4>(xg + 1 * NM * 128, tagc + 1024u, DROP ? p.hm_b + (size_t)t * slot : p.hsb[0] + (size_t)(t + 1) * slot, HD * 2, row0, B, H1, ot, wave, member, local, p.err, 12, &s_dead,
              [&] {
                publish_h(0, hp1);
                if constexpr (DROP) {
#pragma unroll
                  for (int rt = 0; rt < 2; ++rt) {
                    const int row = row0 + 16 * rt + oc16;
                    pst8(oq == 0 && row < B ? (void*)(p.hm_b + (size_t)t * slot + (size_t)row * HD + 16 * member + 4 * wave) : (void*)otrash, hpm[rt], local);
                  }
                }
              }, [&] { store_layer(ot, 0, t, g1, c1, hp1, false); }, gs, p.stamps != nullptr, tl0 ? tl0 + 16 : nullptr);
    if (s_dead) break;
    DC_STAMP(2);
    f32x4 g2[2]; u32x2 hp2[2];
    bf16x8 cav[16];                                                // tile `wave` of the pre-multiplied context of this member's row: lands while layer 2 runs
    {
      const bf16_t* carow = ca + (size_t)min(16 * wave + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
      for (int s = 0; s < 16; ++s) cav[s] = *reinterpret_cast<const bf16x8*>(carow + 32 * s);
    }
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
      gather<6>(xg + 2 * NM * 128, tagc + 2048u, p.hsb[1] + (size_t)(t + 1) * slot, HD * 2, row0, B, H2, ot, wave, member, local, p.err, 13, &s_dead,
                [&] { publish_h(1, hp2); }, [&] { store_layer(ot, 1, t, g2, c2, hp2, true); }, gs, p.stamps != nullptr, tl0 ? tl0 + 32 : nullptr);
      if (s_dead) break;
      DC_STAMP(4);
      if constexpr (!DEC) {                                        // zx1 of the next step: lands while the attention runs
        const int tn = min(t + 1, L - 1);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const int row = min(row0 + 16 * rt + oc16, B - 1);
          const size_t zr = p.zx_tok ? (size_t)(min(max(ntok[rt], 1), p.V) - 1) : (size_t)tn * B + row;
#pragma unroll
          for (int i = 0; i < 4; ++i) zxr[rt][i] = p.zx1[zr * 4 * HD + i * HD + ounit];
          if (p.zx_tok) ntok[rt] = p.zx_tok[(int64_t)min(t + 2, L - 1) * p.zx_st + (int64_t)row * p.zx_sb];
        }
      }
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
      for (int i = 0; i < 16; ++i) cv[i] = *reinterpret_cast<const bf16x8*>(cx + (size_t)min(4 * i + wave, T - 1) * HD + 8 * olane);
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
      gather<1>(xg + 3 * NM * 128, tagc + 3072u, p.cat_b + (size_t)t * B * 2 * HD, HD * 4, row0, B, F, ot, wave, member, local, p.err, 14, &s_dead,
                [&] { pst4(rvalid ? (void*)(p.cat_b + ((size_t)t * B + arow) * 2 * HD + 2 * ot) : (void*)otrash, cb, local); },      // c of row `member`: units 2 tid, 2 tid + 1
                [&] { st4f(rvalid && ot < T ? (void*)(p.a_all + ((size_t)t * B + arow) * T + ot) : (void*)otrash, av); }, gs, p.stamps != nullptr, tl0 ? tl0 + 48 : nullptr);
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
        if constexpr (DROP) {                                      // Dropout on the attention output (LSTM.lua:116-118): idx = row * Hd + column, step offset t * B * Hd
          DropSpec d = p.drop_out; d.off = (long long)t * (long long)slot;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] *= d.mask((long long)(row0 + 16 * wave + c16) * HD + 16 * member + 4 * q + i);
        }
        ov = v; ovb = u32x2{bfpair(v[0], v[1]), bfpair(v[2], v[3])};         // published behind the polls of the next step's first gather
        if constexpr (DEC) *reinterpret_cast<f32x4*>(outs + (16 * wave + c16) * 16 + 4 * q) = v;
      }
    }
    if constexpr (DEC) {
      // projector (output_projector.lua:3-8) on fp32 out: this member's 16 units against its slice of W_o, 32 rows x V partial logits
      lds_barrier();
      float* const pp = p.pbuf + (size_t)group * 32 * 32 * 40;                   // [row][source member][40]
      raise_and_wait(xg + 4 * NM * 128, tagc + 512u, ot, wave, member, local, p.err, 16, &s_dead, [&] {
        const int r = ot >> 3, j = ot & 7;
        float o16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) o16[u] = outs[r * 16 + u];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int v = j + 8 * k;
          float acc = 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u) acc = fmaf(wos[v * 16 + u], o16[u], acc);
          pst4(pp + ((size_t)r * 32 + member) * 40 + v, __builtin_bit_cast(unsigned, acc), local);
        }
      });
      if (s_dead) break;
      if (wave == 0) {                     // LogSoftMax + selection of this member's row (project_select_kernel at beam 1)
        const int V = p.V;
        float x = -INFINITY;
        if (olane < V) {
          x = p.bo[olane];
          float* const base = pp + (size_t)member * 32 * 40 + olane;
          float pv[32];
#pragma unroll
          for (int sm = 0; sm < 32; ++sm)
            pv[sm] = local ? __hip_atomic_load(base + sm * 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(base + sm * 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
          for (int sm = 0; sm < 32; ++sm) x += pv[sm];
        }
        const float mx = wave_reduce(x, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
        const float sum = wave_reduce(olane < V ? expf(x - mx) : 0.f, 0.f, [](float a, float b) { return a + b; });
        float lp = olane < V ? x - (mx + logf(sum)) : -INFINITY;
        if (t > 0) {
          if (olane == 0 && (prev_tok == 1 || prev_tok == 3)) lp = 0.f;          // model.lua:448-449: after PAD / EOS only PAD, at no cost
          lp += score;                                                          // model.lua:450
        }
        unsigned long long tmask = 0;                                            // -use_dictionary: only tokens that continue the row's trie node (model.lua:413,469)
        if (p.trie_mask) {
          tmask = p.trie_mask[node];
          const bool ok = (t > 0 && olane == 0) || ((tmask >> olane) & 1ull);      // PAD is always admissible after the first step
          if (olane < V && !ok) lp = -INFINITY;
        }
        const float best = wave_reduce(lp, -INFINITY, [](float a, float b) { return fmaxf(a, b); });
        const unsigned long long tie = __ballot(lp == best && olane < V);        // descending score, ties -> lowest index
        const int bi = tie ? __ffsll((long long)tie) - 1 : 0;
        score = best; prev_tok = bi + 1;
        if (p.trie_mask && !(t > 0 && bi == 0) && ((tmask >> bi) & 1ull))        // trie_next: PAD keeps the node (model.lua:502-503)
          node = p.trie_child[p.trie_base[node] + __popcll(tmask & ((1ull << bi) - 1ull))];
        if (olane == 0) {
          if (rvalid) { p.labels[(size_t)arow * p.tok0_stride + t] = bi + 1; if (t == L - 1) p.scores[arow] = best; }
          pst4(p.tokx + (size_t)group * 32 + member, (unsigned)(bi + 1), local);
        }
      }
    }
    DC_STAMP(7);
  }
  if constexpr (DEC) {
    if (t_exit >= 0 && wave == 0 && lane == 0 && rvalid) {
      for (int tt = t_exit; tt < L; ++tt) p.labels[(size_t)arow * p.tok0_stride + tt] = 1;
      p.scores[arow] = score;
    }
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

// =============================================================================================================================
// Decoder BPTT (model.lua:643-661, t = L..1; cell backward LSTM.lua:79-105 through nngraph, attention LSTM.lua:124-162) in ONE launch,
// same groups as the forward kernel: member m owns units 16m .. 16m+15 of every 512-wide vector of its 32 rows.  Per step, five
// exchanges inside the group (flags + the real output tensors as payload, as in the forward kernel):
//   d pre = (d out_proj(t) + d feed) (1 - out^2)                         own units                      -> all-gather (bf16)
//   [d c ; d h2a] = d pre W_c                                            own columns                    -> row r's d c to member r (fp32)
//   member r: d a = ctx d c, d s = a (d a - a . d a), d q = d s ctx       the row's attention backward   -> all-gather d q (bf16)
//   d h2 = d q W_a + d h2a + d h2rec;  cell backward of layer 2          own units: d z2 (4 gates)      -> all-gather d z2 (bf16, 4 KB rows)
//   d h1 = d z2 W2_i2h + d h1rec;      cell backward of layer 1          own units: d z1                -> all-gather d z1
//   d h2rec = d z2 W2_h2h,  d h1rec = d z1 W1_h2h,  d feed = d z1 W1_i2h[:, E:]                           own units, for step t-1
// The four K = 2048 weight slices (rows 16m .. 16m+15 of the TRANSPOSED matrices, K split over the four waves) stay in registers:
// 256 VGPRs per lane; W_c^T / W_a^T slices (48) too.  Everything the hoisted weight-gradient GEMMs and attention_dctx read is
// written in the launch chain's layouts: d pre, d c (in d cat), d s, d q, d z (fp32 + bf16), plus the final d c / d h of both layers.
constexpr int BWD_LDS_BYTES = R * PZ + 16384 + 4096;               // operand (d pre / d q alias its start) + partial tiles + attention scratch

template <bool DROP = false>                                        // DROP: the forward kernel ran with nn.Dropout(p > 0) (masked attention output, masked layer-2 input)
__global__ __launch_bounds__(256, 1) void dec_cl_bwd_kernel(DecClBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const X = lds;                                     // [32][PA]: d pre, later d q;  [32][PZ]: d z2, later d z1
  float* const red = reinterpret_cast<float*>(lds + R * PZ);        // [4 waves][2 tiles][2 rt][64 lanes][4]: K-split partial tiles (16 KB)
  float* const part = red;                                          // attention: [4 waves][512] partial d q
  float* const da = reinterpret_cast<float*>(lds + R * PZ + 16384); // [256] d a
  bf16_t* const dchl = reinterpret_cast<bf16_t*>(lds + R * PZ + 16384 + 1024);   // [2][512]: the row's d c as hi + lo bf16 (B operands)
  __shared__ int s_local, s_dead;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % NM, gl = (i8 / NM) * 8 + xcd;
  if (gl >= p.ngroups) return;
  const int group = p.group0 + gl;
  const int B = p.B, T = p.T, L = p.L, row0 = group * R;
  const size_t slot = (size_t)B * HD;

  u64* const xt = p.xtab + (size_t)group * NM;
  if (tid == 0) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 15u;
    stg64(xt + member, ((u64)p.epoch << 32) | (u64)(xcc + 1u));
    int same = 1;
    for (int m = 0; m < NM; ++m) {
      u64 v; int spins = 0;
      while ((unsigned)((v = ldg64(xt + m)) >> 32) != p.epoch) { if (++spins > DC_SPIN_LIMIT) { atomicExch(p.err, 25); same = 0; break; } __builtin_amdgcn_s_sleep(2); }
      if ((unsigned)v != xcc + 1u) same = 0;
    }
    s_local = same && !p.force_remote; s_dead = 0;
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
      wct[0][s] = *reinterpret_cast<const bf16x8*>(p.wc_t + (size_t)(16 * member + c16) * HD + 128 * wave + 32 * s + 8 * q);          // d c columns
      wct[1][s] = *reinterpret_cast<const bf16x8*>(p.wc_t + (size_t)(HD + 16 * member + c16) * HD + 128 * wave + 32 * s + 8 * q);     // d h_top columns
      wat[s] = *reinterpret_cast<const bf16x8*>(p.wa_t + (size_t)(16 * member + c16) * HD + 128 * wave + 32 * s + 8 * q);
    }
  }
  unsigned* const xg = reinterpret_cast<unsigned*>(p.xbuf) + (size_t)group * 5 * NM * 128;      // flags [kind][reader][member][wave]
  const int arow = row0 + member; const bool rvalid = arow < B;
  const bf16_t* const cx = p.ctxb + (size_t)min(arow, B - 1) * T * HD;
  const int ntile = (T + 15) >> 4;
  // elementwise layout (waves 0, 1 = row tile): lane -> row 16 wave + c16, units 16 member + 4q .. + 3
  f32x4 dc1 = {0.f, 0.f, 0.f, 0.f}, dc2 = dc1, dh1rec = dc1, dh2rec = dc1, dfeed = dc1;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  [[maybe_unused]] u64 stamp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();

  // K-split product of NT tiles: partial tiles -> LDS -> waves 0, 1 hold (row tile = wave) the sums
  auto reduce_tiles = [&](auto& acc, auto& v) {
    constexpr int NT = sizeof(v) / sizeof(v[0]);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) *reinterpret_cast<f32x4*>(red + ((size_t)((wave * 2 + n) * 2 + rt) * 64 + lane) * 4) = acc[n][rt];
    lds_barrier();
    if (wave < 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        v[n] = *reinterpret_cast<const f32x4*>(red + ((size_t)((0 * 2 + n) * 2 + wave) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) v[n] += *reinterpret_cast<const f32x4*>(red + ((size_t)((w * 2 + n) * 2 + wave) * 64 + lane) * 4);
      }
    }
  };
  // cell backward of one layer on this lane's 4 units (EpGatesBwd): d h -> d z (4 gates x 4 units), d c state update (waves 0, 1).
  // (Loading the saved gates / cell states a phase ahead was measured slower: 24 more live registers spill.)
  struct CellIn { f32x4 cn, cp, g[4]; };
  auto cell_load = [&](int ot, int l, int t, CellIn& ci) {
    if (wave < 2) {
      const int c16_ = ot & 15, q_ = (ot >> 4) & 3;
      const int row = min(row0 + 16 * wave + c16_, B - 1), u0 = 16 * member + 4 * q_;
      ci.cn = *reinterpret_cast<const f32x4*>(p.cs[l] + (size_t)(t + 1) * slot + (size_t)row * HD + u0);
      ci.cp = *reinterpret_cast<const f32x4*>(p.cs[l] + (size_t)t * slot + (size_t)row * HD + u0);
#pragma unroll
      for (int i = 0; i < 4; ++i) ci.g[i] = *reinterpret_cast<const f32x4*>(p.gates[l] + (((size_t)t * B + row) * HD + u0 + i) * 4);
    }
  };
  auto cell_bwd = [&](const CellIn& ci, const f32x4& dh, f32x4& dcs, f32x4 (&dz)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ig = ci.g[i][0], fg = ci.g[i][1], og = ci.g[i][2], gg = ci.g[i][3];
      const float tc = tanhf_(ci.cn[i]);
      const float dcv = dh[i] * og * (1.f - tc * tc) + dcs[i];
      const float d_o = dh[i] * tc, di = dcv * gg, dg = dcv * ig, df = dcv * ci.cp[i];
      dz[0][i] = di * ig * (1.f - ig); dz[1][i] = df * fg * (1.f - fg); dz[2][i] = d_o * og * (1.f - og); dz[3][i] = dg * (1.f - gg * gg);
      dcs[i] = dcv * fg;
    }
  };
  // d out_proj(t) and out(t) of this lane's units: loaded one step ahead
  f32x4 dpn = {0.f, 0.f, 0.f, 0.f}, on = dpn;
  auto load_step_inputs = [&](int ot, int t) {
    if (wave < 2) {
      const size_t eo = (size_t)min(row0 + 16 * wave + (ot & 15), B - 1) * HD + 16 * member + 4 * ((ot >> 4) & 3);
      dpn = *reinterpret_cast<const f32x4*>(p.dout_proj + (size_t)t * slot + eo);
      on = *reinterpret_cast<const f32x4*>(p.out + (size_t)(t + 1) * slot + eo);
    }
  };
  load_step_inputs(tid, L - 1);
  for (int t = L - 1; t >= 0; --t) {
    const unsigned tagc = p.epoch * 4096u + (unsigned)(t + 1);     // flags of step t: + kind * 512
    int ot = tid; asm volatile("" : "+v"(ot));                      // opaque per-step copy of the thread id (see the forward kernel)
    const int olane = ot & 63, oc16 = ot & 15, oq = (ot >> 4) & 3;
    unsigned char* const otrash = reinterpret_cast<unsigned char*>(p.err + 16) + ot * 16;
    const int erow = row0 + 16 * wave + oc16;                        // elementwise layout (waves 0, 1)
    const bool eok = wave < 2 && erow < B;
    const size_t eoff = (size_t)min(erow, B - 1) * HD + 16 * member + 4 * oq;
    // =================== d pre
    f32x4 dpre = {0.f, 0.f, 0.f, 0.f};
    if (wave < 2) {
      if constexpr (DROP) {                                         // `out` holds Dropout(tanh): the gradient passes the mask, tanh' needs the unmasked value (dpre_kernel)
        DropSpec d = p.drop_out; d.off = (long long)t * (long long)slot;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mk = d.mask((long long)erow * HD + 16 * member + 4 * oq + i);
          const float o = mk != 0.f ? on[i] / mk : 0.f;
          dpre[i] = (dpn[i] + dfeed[i]) * mk * (1.f - o * o);
        }
      } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) dpre[i] = (dpn[i] + dfeed[i]) * (1.f - on[i] * on[i]);
      }
    }
    raise_and_wait(xg + 0 * NM * 128, tagc, ot, wave, member, local, p.err, 21, &s_dead,
                   [&] { pst8(eok ? (void*)(p.dpre_b + (size_t)t * slot + eoff) : (void*)otrash, u32x2{bfpair(dpre[0], dpre[1]), bfpair(dpre[2], dpre[3])}, local); });
    DC_STAMP(0);
    fetch_rows<1, 1>(p.dpre_b + (size_t)t * slot, HD * 2, row0, B, X, PA, ot, member, local,
                     [&] { st16f(eok ? (void*)(p.dpre + (size_t)t * slot + eoff) : (void*)otrash, dpre); });
    DC_STAMP(1);
    // attention prefetches of this member's row (independent of the step's gradients): a(t), the d a tile of ctx
    float aj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) aj[j] = (olane + 64 * j < T) ? p.a_all[((size_t)t * B + min(arow, B - 1)) * T + olane + 64 * j] : 0.f;
    // the tile of ctx for d a (rows 16 wave + c16): in flight across the d cat product and the d c exchange
    bf16x8 cxv[16];
    {
      const bf16_t* r1 = cx + (size_t)min(16 * wave + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
      for (int s = 0; s < 16; ++s) cxv[s] = *reinterpret_cast<const bf16x8*>(r1 + 32 * s);
    }
    if (s_dead) break;
    // =================== [d c ; d h2a] = d pre W_c  (two tiles of 16 columns)
    f32x4 dcat[2];
    {
      f32x4 acc[2][2];
#pragma unroll
      for (int n = 0; n < 2; ++n) { acc[n][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[n][1] = acc[n][0]; }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(X + (size_t)(16 * rt + c16) * PA + (128 * wave + 32 * s + 8 * q) * 2);
          acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wct[0][s], bv, acc[0][rt], 0, 0, 0);
          acc[1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wct[1][s], bv, acc[1][rt], 0, 0, 0);
        }
      reduce_tiles(acc, dcat);
    }
    DC_STAMP(2);
    // d c of all rows -> the rows' owners (payload: the c half of d cat, fp32; attention_dctx reads it after the loop)
    raise_and_wait(xg + 1 * NM * 128, tagc + 512u, ot, wave, member, local, p.err, 22, &s_dead,
                   [&] { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(eok ? (void*)(p.dcat + ((size_t)t * B + min(erow, B - 1)) * 2 * HD + 16 * member + 4 * oq) : (void*)otrash), "v"(dcat[0]) : "memory"); });
    DC_STAMP(3);
    if (s_dead) break;
    {                                                                // this member's row: 512 floats -> hi + lo bf16 in LDS
      unsigned lo, hi;
      u64 v; asm volatile("global_load_dwordx2 %0, %1, %2 sc1" : "=v"(v) : "v"((unsigned)(ot * 8)), "s"(p.dcat + ((size_t)t * B + min(arow, B - 1)) * 2 * HD) : "memory");
      wait_vm<0>();
      asm volatile("" : "+v"(v));
      const float x0 = __builtin_bit_cast(float, (unsigned)v), x1 = __builtin_bit_cast(float, (unsigned)(v >> 32));
      const float h0 = (float)(bf16_t)x0, h1 = (float)(bf16_t)x1;
      hi = bfpair(h0, h1); lo = bfpair(x0 - h0, x1 - h1);
      reinterpret_cast<unsigned*>(dchl)[ot] = hi; reinterpret_cast<unsigned*>(dchl + HD)[ot] = lo;
      lds_barrier();
    }
    // =================== attention backward of row `member`
    unsigned dqb; float dsv_own; f32x2 dqv;
    {
      const unsigned char* hrow = reinterpret_cast<const unsigned char*>(dchl) + 16 * q;
      for (int tile = wave; tile < ntile; tile += 4) {
        if (tile != wave) {
          const bf16_t* r2 = cx + (size_t)min(16 * tile + oc16, T - 1) * HD + 8 * oq;
#pragma unroll
          for (int s = 0; s < 16; ++s) cxv[s] = *reinterpret_cast<const bf16x8*>(r2 + 32 * s);
        }
        f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cxv[s], *reinterpret_cast<const bf16x8*>(hrow + 64 * s), a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cxv[s], *reinterpret_cast<const bf16x8*>(hrow + HD * 2 + 64 * s), a1, 0, 0, 0);
        }
        if (c16 == 0) *reinterpret_cast<f32x4*>(da + 16 * tile + 4 * q) = a0 + a1;
      }
      bf16x8 cv[16];                                               // second pass over ctx (d q): in flight across the d s arithmetic
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
      dsv_own = wave == 0 ? dsj[0] : wave == 1 ? dsj[1] : wave == 2 ? dsj[2] : dsj[3];      // d s[tid]
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
          const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dsj[c]), 4 * i + wave));     // 0 beyond T
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
      dqv = f32x2{v0, v1}; dqb = bfpair(v0, v1);
    }
    DC_STAMP(4);
    raise_and_wait(xg + 2 * NM * 128, tagc + 1024u, ot, wave, member, local, p.err, 23, &s_dead,
                   [&] { pst4(rvalid ? (void*)(p.dq_b + (size_t)t * slot + (size_t)arow * HD + 2 * ot) : (void*)otrash, dqb, local); });
    DC_STAMP(5);
    fetch_rows<1, 2>(p.dq_b + (size_t)t * slot, HD * 2, row0, B, X, PA, ot, member, local,
                     [&] {
                       st4f(rvalid && ot < T ? (void*)(p.ds_all + ((size_t)t * B + arow) * T + ot) : (void*)otrash, dsv_own);
                       asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(rvalid ? (void*)(p.dq + (size_t)t * slot + (size_t)arow * HD + 2 * ot) : (void*)otrash), "v"(dqv) : "memory");
                     });
    if (s_dead) break;
    DC_STAMP(6);
    // =================== d h2 = d q W_a + d h2a + d h2rec;  cell backward of layer 2
    f32x4 dz[4];
    {
      f32x4 acc[1][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}}, v[1];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wat[s], *reinterpret_cast<const bf16x8*>(X + (size_t)(16 * rt + c16) * PA + (128 * wave + 32 * s + 8 * q) * 2), acc[0][rt], 0, 0, 0);
      reduce_tiles(acc, v);
      if (wave < 2) { const f32x4 dh2 = v[0] + dcat[1] + dh2rec; CellIn ci; cell_load(ot, 1, t, ci); cell_bwd(ci, dh2, dc2, dz); }
    }
    auto dz_payload = [&](int l) {                                   // d z of (row, 4 units) x 4 gates, bf16, [row][gate * 512 + unit]
#pragma unroll
      for (int g = 0; g < 4; ++g)
        pst8(eok ? (void*)(p.dzb[l] + ((size_t)t * B + min(erow, B - 1)) * 4 * HD + g * HD + 16 * member + 4 * oq) : (void*)otrash, u32x2{bfpair(dz[g][0], dz[g][1]), bfpair(dz[g][2], dz[g][3])}, local);
    };
    auto dz_deferred = [&](int l) {                                  // the fp32 copy: 4 stores
#pragma unroll
      for (int g = 0; g < 4; ++g) st16f(eok ? (void*)(p.dz[l] + ((size_t)t * B + min(erow, B - 1)) * 4 * HD + g * HD + 16 * member + 4 * oq) : (void*)otrash, dz[g]);
    };
    // K = 2048 products against the gathered d z: two tiles (matrices ka, kb), this wave's quarter of K
    auto zprod = [&](const bf16x8 (&wa_)[16], const bf16x8 (&wb_)[16], f32x4 (&v)[2]) {
      f32x4 acc[2][2];
#pragma unroll
      for (int n = 0; n < 2; ++n) { acc[n][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[n][1] = acc[n][0]; }
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          const bf16x8 bv = *reinterpret_cast<const bf16x8*>(X + (size_t)(16 * rt + c16) * PZ + (512 * wave + 32 * s + 8 * q) * 2);
          acc[0][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa_[s], bv, acc[0][rt], 0, 0, 0);
          acc[1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb_[s], bv, acc[1][rt], 0, 0, 0);
        }
      reduce_tiles(acc, v);
    };
    DC_STAMP(7);
    raise_and_wait(xg + 3 * NM * 128, tagc + 1536u, ot, wave, member, local, p.err, 24, &s_dead, [&] { dz_payload(1); });
    DC_STAMP(8);
    fetch_rows<4, 4>(p.dzb[1] + (size_t)t * B * 4 * HD, HD * 8, row0, B, X, PZ, ot, member, local, [&] { dz_deferred(1); });
    if (s_dead) break;
    DC_STAMP(9);
    {
      f32x4 v[2];
      zprod(wz[0], wz[1], v);                                               // d z2 W2_i2h (-> d h1), d z2 W2_h2h (-> d h2rec of step t-1)
      if (wave < 2) {
        dh2rec = v[1];
        if constexpr (DROP) {                                       // layer 2 read Dropout(h1): its input gradient passes the mask (EpGatesBwd::drop)
          DropSpec d = p.drop_h; d.off = (long long)t * (long long)slot;
#pragma unroll
          for (int i = 0; i < 4; ++i) v[0][i] *= d.mask((long long)erow * HD + 16 * member + 4 * oq + i);
        }
        const f32x4 dh1 = v[0] + dh1rec; CellIn ci; cell_load(ot, 0, t, ci); cell_bwd(ci, dh1, dc1, dz);
      }
    }
    DC_STAMP(10);
    raise_and_wait(xg + 4 * NM * 128, tagc + 2048u, ot, wave, member, local, p.err, 26, &s_dead, [&] { dz_payload(0); });
    DC_STAMP(11);
    fetch_rows<4, 4>(p.dzb[0] + (size_t)t * B * 4 * HD, HD * 8, row0, B, X, PZ, ot, member, local, [&] { dz_deferred(0); });
    if (s_dead) break;
    DC_STAMP(12);
    if (t > 0) load_step_inputs(ot, t - 1);
    {
      f32x4 v[2];
      zprod(wz[2], wz[3], v);                                               // d z1 W1_h2h (-> d h1rec), d z1 W1_i2h[:, E:] (-> d feed)
      if (wave < 2) { dh1rec = v[0]; dfeed = v[1]; }
    }
    lds_barrier();                                                  // the d z operand is dead: the next step overwrites its start
    DC_STAMP(13);
  }
  // d c / d h of the initial decoder state, for the encoder's backward pass (model.lua:662-690)
  if (!s_dead && wave < 2) {
    const int erow = row0 + 16 * wave + c16;
    if (erow < B) {
      const size_t o = (size_t)erow * HD + 16 * member + 4 * q;
      *reinterpret_cast<f32x4*>(p.dc_st[0] + o) = dc1; *reinterpret_cast<f32x4*>(p.dc_st[1] + o) = dc2;
      *reinterpret_cast<f32x4*>(p.dh_rec[0] + o) = dh1rec; *reinterpret_cast<f32x4*>(p.dh_rec[1] + o) = dh2rec;
      *reinterpret_cast<f32x4*>(p.dfeed + o) = dfeed;
    }
  }
  wait_vm<0>();
#ifdef DC_DEBUG_STAMPS
  if (p.stamps && wid == 0 && tid == 0)
    for (int k = 0; k < 16; ++k) p.stamps[k] = stamp[k];
#endif
}

// ---------------------------------------------------------------------------------------------
size_t dec_cluster_xbuf_bytes(int B) { return (size_t)((B + R - 1) / R) * 5 * NM * 128 * sizeof(unsigned) + 256; }
size_t dec_cluster_pbuf_bytes(int B) { return (size_t)((B + R - 1) / R) * (2 * 32 * 32 * 40 + 32) * sizeof(float) + 256; }   // (two step parities for dec_chain.hip)   // partial logits + tokens (greedy decode)
size_t dec_cluster_xtab_bytes(int B) { return (size_t)((B + R - 1) / R) * NM * sizeof(u64) + 512; }   // + two debugging stamp areas
bool dec_cluster_supported(int Hd, int Ld, int input_feed, int T, int L, int cus) { return Hd == HD && Ld == 2 && input_feed && T >= 1 && T <= 256 && L + 2 < 1024 && cus >= 8 * NM; }

size_t dec_cluster_bwd_xbuf_bytes(int B) { return (size_t)((B + R - 1) / R) * 5 * NM * 128 * sizeof(unsigned) + 256; }
bool dec_cluster_bwd_supported(int Hd, int Ld, int input_feed, int T, int L, int cus) { return dec_cluster_supported(Hd, Ld, input_feed, T, L, cus) && L + 2 < 512; }
void dec_cluster_backward(hipStream_t s, const DecClBwdArgs& a0) {
  if (a0.drop_h.thr == 0 && a0.drop_out.thr == 0 && dec_chain_enabled() && !getenv("AOCR_NO_DEC_CHAINS_BWD")) { dec_chain_backward(s, a0); return; }      // round 5: two chains per group, tag-free exchange
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  (void)hipFuncSetAttribute((const void*)dec_cl_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS_BYTES);
  (void)hipFuncSetAttribute((const void*)dec_cl_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)BWD_LDS_BYTES);
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClBwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr;
    a.stamps = getenv("AOCR_DC_STAMPS") ? a.xtab + (size_t)groups * NM + 16 : nullptr;   // debugging aid (-DDC_DEBUG_STAMPS): cycles per phase of workgroup 0
    if (a.drop_h.thr != 0 || a.drop_out.thr != 0) hipLaunchKernelGGL(dec_cl_bwd_kernel<true>, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), (size_t)BWD_LDS_BYTES, s, a);
    else hipLaunchKernelGGL(dec_cl_bwd_kernel<false>, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), (size_t)BWD_LDS_BYTES, s, a);
  }
}

void dec_cluster_forward(hipStream_t s, const DecClFwdArgs& a0, bool greedy_decode) {
  if (a0.drop_h.thr == 0 && a0.drop_out.thr == 0 && dec_chain_enabled() && !(greedy_decode && getenv("AOCR_NO_DEC_CHAINS_GREEDY"))) { dec_chain_forward(s, a0, greedy_decode); return; }      // round 5: two chains per group, tag-free exchange
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  const int groups = (a0.B + R - 1) / R, per_pass = std::max(8, cus / (8 * NM) * 8);
  const size_t lds = LDS_BYTES;
  (void)hipFuncSetAttribute((const void*)dec_cl_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)dec_cl_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute((const void*)dec_cl_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int g0 = 0; g0 < groups; g0 += per_pass) {
    DecClFwdArgs a = a0; a.group0 = g0; a.ngroups = std::min(per_pass, groups - g0); a.force_remote = getenv("AOCR_CL_REMOTE") != nullptr; a.no_early = getenv("AOCR_NO_DEC_EARLY") != nullptr;
    a.stamps = getenv("AOCR_DC_STAMPS") ? a.xtab + (size_t)groups * NM : nullptr;   // debugging aid: cycles per phase of workgroup 0
    if (greedy_decode) { a.tokx = reinterpret_cast<unsigned*>(a.pbuf + (size_t)groups * 32 * 32 * 40); hipLaunchKernelGGL(dec_cl_fwd_kernel<true>, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), lds, s, a); }
    else if (a.drop_h.thr != 0 || a.drop_out.thr != 0) hipLaunchKernelGGL((dec_cl_fwd_kernel<false, true>), dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(dec_cl_fwd_kernel<false>, dim3(8 * NM * ((a.ngroups + 7) / 8)), dim3(256), lds, s, a);
  }
}

}  // namespace aocr
