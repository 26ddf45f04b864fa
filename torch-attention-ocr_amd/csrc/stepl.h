// stepl.h -- recurrent-step products at LARGE batch (round 6: the reference's default shape, 400 rows x Hd = 1024; BASELINE config 5's decoder).
//
// gemm_step_kernel (mfma_gemm.h) is built for latency at 32-256 rows: 32-row tiles, every wave stages its own K quarter through registers into a private
// LDS image.  At 400 rows a gate launch took 34-40 us for 6.9 GFLOP (0.07 of the MFMA peak; profiles/r05_ref_step_trace.txt).  Measured with
// tools/ubench/step400.hip (the library's own kernels and epilogue on that shape): 13-28 us of it was the epilogue's "prefetch" -- every call consumed its loads at
// once (bias sum, `+= zx`), i.e. 8-16 serial memory round trips per thread in front of the K loop (fixed in epilogues.h for every step kernel); the K loop
// itself ran at 21 us, bound by what one wave per SIMD can issue: staging loads, LDS writes, fragment reads and MFMAs of a wave serialise.  Here:
//   * operands staged by LDS-DMA (global_load_lds_dwordx4, whole 128-byte lines: a piece = 8 rows x 128 B) into a ring of 64-deep sub-stages SHARED by the
//     workgroup -- no staging registers, no ds_write pass, NS - SUBS sub-stages in flight per CU;
//   * EIGHT waves: four K slices x two halves of the tile's columns (or rows), two waves per SIMD, so one wave's DMA issue (~80 cycles per instruction) runs
//     under the other's MFMAs; a wave multiplies its half tile over its k slice of every sub-stage, so an A / B fragment read from LDS feeds 2 MFMAs;
//   * the four K partials meet in LDS (the ring, reused) and the gate epilogue runs on FOUR consecutive hidden units per thread: 16-byte loads of zx / c_prev and
//     16-byte stores of c, h, the saved gates (8-byte for the bf16 shadows) -- a quarter of the scalar epilogue's memory instructions;
//   * workgroups renumbered so that an XCD's workgroups share few column blocks (its L2 holds their weights once) x all row blocks.
// Measured and dropped: split K over workgroups with the partial tiles meeting in memory (sc1 stores, counter, the last workgroup sums: ~8 us of dependent
// round trips -- store acknowledgement, atomic, partner loads, epilogue operands -- on kernels of 10-20 us: the plain N = Hd products of a step stay on
// gemm_step_kernel's 416 small workgroups, 16.4 us at K = 4096 against 17-20 here), padded row strides and a per-workgroup rotation of the K loop (no L2 channel effect: same time), 128 x 64 tiles on half gate tiles (256
// workgroups, a quarter of them on the 16 rows past 384: slower than 64 x 128 on 224).
// The fp32 sums meet in a different order than in gemm_step_kernel: results agree to summation-order noise, not bit for bit.
// Same argument block (SmallArgs2) and epilogue objects as gemm_step_kernel, so it drops into launch_small_bf16_hh.
#pragma once
#include "mfma_gemm.h"

namespace aocr {

template <int MT, int NT> constexpr int stepl_sub_bytes() { return (MT + NT) * 32 * 128; }
template <int MT, int NT, int NS> constexpr int stepl_lds_bytes() { return NS * stepl_sub_bytes<MT, NT>(); }

// GATES: 0 plain (NT column tiles of 32, one elem<1> call per tile and element), 1 gate tiles (NT = 4: tile = gate, 32 hidden units per workgroup; epilogue on
// four units per thread when the epilogue object says its pointers allow it, EP::cell4_ok).  NW = 4 or 8 waves; SPLITN: the two wave halves split the column
// tiles (else the row tiles).  NS sub-stages of 64 k in the ring, SUBS consumed per barrier.
template <int MT, int NT, int GATES, class EP, int NS, int SUBS, int NW, bool SPLITN>
__global__ __launch_bounds__(64 * NW, 1) void gemm_stepl_kernel(SmallArgs2<LoadKh2, LoadKh2, EP> zz, int gate_stride, int gx, int gy) {
  constexpr int SUB = stepl_sub_bytes<MT, NT>(), NG = NS / SUBS, NH = NW / 4;
  constexpr int PPW = (MT + NT) * 4 / NW, APW = MT * 4 / NW;                       // pieces per wave and sub-stage; the first APW of them are A pieces
  constexpr int MTW = SPLITN ? MT : MT / NH, NTW = SPLITN ? NT / NH : NT;         // tiles of one wave
  static_assert(NW == 4 || NW == 8, "four K slices x one or two halves");
  static_assert(NS % SUBS == 0 && NG >= 2, "ring = whole groups");
  static_assert((NG - 2) * SUBS * PPW <= 63, "vmcnt field");
  static_assert(4 * MT * NT * 4096 <= NS * SUB, "the reduction image reuses the ring");
  static_assert((MT * 4) % NW == 0 && (NT * 4) % NW == 0 && (SPLITN ? NT : MT) % NH == 0, "whole pieces / tiles per wave");
  static_assert(GATES == 0 || (GATES == 1 && NT == 4), "gate tiles: tile = gate");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * SUB];         // the ONLY LDS object
  // XCD-aware renumbering: consecutive ids share an XCD; column-block-major, so an XCD holds few column blocks x all row blocks
  const int nwg = gx * gy, orig = blockIdx.x % nwg, zi = blockIdx.x / nwg;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int bx = bid / gy, by = bid % gy;
  const SmallArgs<LoadKh2, LoadKh2, EP>& g = zz.z[zi];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kw = wave & 3, hf = wave >> 2;                                         // K slice of a sub-stage; half of the tile
  const int r = lane & 31, h = lane >> 5;
  const int m0 = by * 32 * MT;
  const int n0 = GATES ? bx * 32 : bx * 32 * NT;
  const int K = g.K, K0 = g.a.K0;
  // the operand descriptors as scalars of this kernel (read through the argument block they were re-loaded from the kernarg segment, behind branches, per sub-stage)
  const bf16_t* const ap0 = g.a.p0; const bf16_t* const ap1 = g.a.p1; const bf16_t* const bp0 = g.b.p0; const bf16_t* const bp1 = g.b.p1;
  const int alda0 = (int)g.a.ld0, alda1 = (int)g.a.ld1, bldb0 = (int)g.b.ld0, bldb1 = (int)g.b.ld1;
  const int ngrp = (K >> 6) / SUBS;

  // staging roles: piece = 8 rows x 128 B; lane -> row rr = lane >> 3, LDS position p = lane & 7 holding global chunk p ^ swz(row in its 32-row tile);
  // wave w stages pieces w + NW j: rows 8 (w + NW j) + rr of the A tiles (j < APW), then of the B tiles -- row 8 (w & 3) + rr of a 32-row tile either way
  const int rr = lane >> 3, p = lane & 7;
  const int gchunk8 = 8 * (p ^ ((4 * kw + (rr >> 1)) & 7));                        // swz = (row in tile >> 1) & 7; in elements
  int prow[PPW];                                                                  // this lane's global row per piece (plain row numbers: per-segment offset arrays selected by the segment were demoted to scratch)
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    if (j < APW) prow[j] = min(m0 + 8 * (wave + NW * j) + rr, g.a.rows - 1);       // rows past the end: any valid row, result dropped
    else {
      const int q = 8 * (wave + NW * (j - APW)) + rr, ni = q >> 5, qi = q & 31;     // row q of the NT x 32 B rows: tile ni, row qi
      prow[j] = GATES ? ni * gate_stride + n0 + qi : n0 + 32 * ni + qi;
    }
  }
  unsigned char* const wbase = lds + wave * 1024;                                  // (the DMA adds the lane's 16 bytes itself)
  int igrp = 0, islot = 0;                                                         // next group to issue, its ring slot
  // one group = SUBS sub-stages; past the end the last group is fetched again into the slot of a consumed group (never read: the instruction count per group is
  // what the waits assume; row tiles past the end of A re-read row M-1 for the same reason)
#define AOCR_STEPL_ISSUE()                                                                                                              \
  {                                                                                                                                     \
    const int gk = min(igrp, ngrp - 1);                                                                                                 \
    _Pragma("unroll") for (int u = 0; u < SUBS; ++u) {                                                                                   \
      const int k = (gk * SUBS + u) << 6;                                                                                               \
      const bool s1 = k >= K0;                                                                                                          \
      const int kk = s1 ? k - K0 : k;                                                                                                   \
      const bf16_t* const pa = (s1 ? ap1 : ap0) + kk; const bf16_t* const pb = (s1 ? bp1 : bp0) + kk;                                     \
      const int lda = s1 ? alda1 : alda0, ldb = s1 ? bldb1 : bldb0;                                                                     \
      unsigned char* const dst = wbase + (islot * SUBS + u) * SUB;                                                                      \
      _Pragma("unroll") for (int j = 0; j < PPW; ++j) {                                                                                  \
        if (j < APW) dma16(pa + (prow[j] * lda + gchunk8), dst + j * (1024 * NW));                                                       \
        else dma16(pb + (prow[j] * ldb + gchunk8), dst + MT * 4096 + (j - APW) * (1024 * NW));                                           \
      }                                                                                                                                 \
    }                                                                                                                                   \
    ++igrp; islot = islot == NG - 1 ? 0 : islot + 1;                                                                                    \
  }

  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // the epilogue's own operands (zx, c_prev, gates, ...) are requested now so that they arrive during the K loop (loads only: EpGatesFwd::prefetch)
  constexpr bool CELL4 = GATES == 1 && EP::kCell4;
  const bool cell4 = CELL4 && g.ep.cell4_ok();
  // (a) four units per thread: item = (row, 4 columns) of the MT*32 x 32 cell tile, MT*256 items over 64 NW threads
  constexpr int IT4 = CELL4 ? (MT * 256 + 64 * NW - 1) / (64 * NW) : 1;
  typename EP::Pre4 pre4[IT4];
  // (b) one unit per thread and call: accumulator index i = E*kw + e of a 32x32 tile sits in row 8*(i/4) + 4h + i%4; the two halves share the tiles of a K slice
  constexpr int E = 4, TPW = MT * NT / NH;                                        // tiles finished per wave: tile ids hf * TPW ...
  constexpr int NP = GATES ? MT / NH : TPW;                                       // prefetch sets per thread: (row tile) for gate tiles, (tile) otherwise
  typename EP::Pre pre[NP][E];
  const int erow = 8 * ((E * kw) >> 2) + 4 * h + ((E * kw) & 3);                   // first of this thread's E rows within a tile
#define AOCR_STEPL_PREFETCH()                                                                                                           \
  if (cell4) {                                                                                                                               \
    if constexpr (CELL4) {                                                                                                                   \
  _Pragma("unroll")                                                                                                                          \
      for (int i = 0; i < IT4; ++i) { const int it = tid + 64 * NW * i; pre4[i] = g.ep.prefetch4(m0 + (it >> 3), n0 + 4 * (it & 7)); }       \
  _Pragma("unroll")                                                                                                                          \
      for (int i = 0; i < IT4; ++i) { const int it = tid + 64 * NW * i; g.ep.prefetch4_zx(pre4[i], m0 + (it >> 3), n0 + 4 * (it & 7)); }     \
    }                                                                                                                                        \
  } else {                                                                                                                                   \
  _Pragma("unroll")                                                                                                                          \
    for (int q = 0; q < NP; ++q)                                                                                                             \
  _Pragma("unroll")                                                                                                                          \
      for (int e = 0; e < E; ++e) {                                                                                                          \
        const int mt = GATES ? hf * NP + q : (hf * TPW + q) / NT, ni = GATES ? 0 : (hf * TPW + q) % NT;                                      \
        pre[q][e] = g.ep.prefetch(m0 + 32 * mt + erow + e, n0 + 32 * ni + r);                                                                \
      }                                                                                                                                      \
  _Pragma("unroll")                                                                                                                          \
    for (int q = 0; q < NP; ++q)                                                                                                             \
  _Pragma("unroll")                                                                                                                          \
      for (int e = 0; e < E; ++e) {                                                                                                          \
        const int mt = GATES ? hf * NP + q : (hf * TPW + q) / NT, ni = GATES ? 0 : (hf * TPW + q) % NT;                                      \
        g.ep.prefetch_zx(pre[q][e], m0 + 32 * mt + erow + e, n0 + 32 * ni + r);                                                              \
      }                                                                                                                                      \
  }                                                                                                                                         
  AOCR_STEPL_PREFETCH()
#undef AOCR_STEPL_PREFETCH

#pragma unroll
  for (int i = 0; i < NG - 1; ++i) AOCR_STEPL_ISSUE()
  // fragment offsets inside a sub-stage: this wave's k slice kw of the 64-deep sub-stage, its half of the tiles
  const int mt0 = SPLITN ? 0 : hf * MTW, nt0 = SPLITN ? hf * NTW : 0;
  const unsigned foff = (unsigned)(r * 128 + (((2 * kw + h) ^ ((r >> 1) & 7)) << 4));
  int rslot = 0;
  for (int gi = 0; gi < ngrp; ++gi) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NG - 2) * SUBS * PPW) : "memory");   // this wave's pieces of group gi have landed (NG - 2 groups stay in flight)
    __builtin_amdgcn_s_barrier();                       // ... everyone's have, and everyone is done reading group gi-1
    const unsigned char* const L = lds + rslot * SUBS * SUB + foff;
    rslot = rslot == NG - 1 ? 0 : rslot + 1;
    bf16x8 af[SUBS][MTW], bfr[SUBS][NTW];
#pragma unroll
    for (int u = 0; u < SUBS; ++u) {
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt) af[u][mt] = *reinterpret_cast<const bf16x8*>(L + u * SUB + (mt0 + mt) * 4096);
#pragma unroll
      for (int ni = 0; ni < NTW; ++ni) bfr[u][ni] = *reinterpret_cast<const bf16x8*>(L + u * SUB + MT * 4096 + (nt0 + ni) * 4096);
    }
    AOCR_STEPL_ISSUE()                                  // group gi + NG - 1 -> the slot of group gi-1
#pragma unroll
    for (int u = 0; u < SUBS; ++u)
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mt][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[u][mt], bfr[u][ni], acc[mt][ni], 0, 0, 0);     // (row tiles past the end multiply row M-1: dropped by the epilogue)
  }
#undef AOCR_STEPL_ISSUE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing pieces must land before the ring becomes the reduction image
  __builtin_amdgcn_s_barrier();
  // cross-wave reduction through LDS: the wave of K slice kw parks its tiles at [kw][tile = mt * NT + ni][16][64 lanes]
  float* const red = reinterpret_cast<float*>(lds);
  constexpr int WSTRIDE = MT * NT * 1024;               // floats per K slice
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) red[kw * WSTRIDE + (((mt0 + mt) * NT + nt0 + ni) * 16 + e) * 64 + lane] = acc[mt][ni][e];
  __syncthreads();
  // ---- this thread's sums of the four K slices, as 16-byte units: (a) item i, gate -> vv[4 i + gate] = four consecutive columns; (b) set q, tile ni -> vv[.] = its E = 4 rows
  constexpr int NVA = CELL4 ? IT4 * 4 : 0, NVB = GATES ? NP * NT : NP, NVM = NVA > NVB ? NVA : NVB;
  f32x4 vv[NVM];
  if (cell4) {
    if constexpr (CELL4) {
      // element (row, col) of tile t sits at [t][i = 4 (row >> 3) + (row & 3)][lane = 32 ((row >> 2) & 1) + col]: four consecutive columns are 16 contiguous bytes
#pragma unroll
      for (int i = 0; i < IT4; ++i) {
        const int it = min(tid + 64 * NW * i, MT * 256 - 1);
        const int row = it >> 3, c0 = 4 * (it & 7), mt = row >> 5, rt = row & 31;
        const int base = (mt * NT * 16 + 4 * (rt >> 3) + (rt & 3)) * 64 + 32 * ((rt >> 2) & 1) + c0;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) {
          const float* const q = red + base + gate * 1024;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(q), s1 = *reinterpret_cast<const f32x4*>(q + WSTRIDE), s2 = *reinterpret_cast<const f32x4*>(q + 2 * WSTRIDE),
                      s3 = *reinterpret_cast<const f32x4*>(q + 3 * WSTRIDE);
          vv[4 * i + gate] = (s0 + s1) + (s2 + s3);
        }
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int mt = GATES ? hf * NP + q : (hf * TPW + q) / NT, ni0 = GATES ? 0 : (hf * TPW + q) % NT;
#pragma unroll
      for (int ni = 0; ni < (GATES ? NT : 1); ++ni)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int idx = ((mt * NT + ni0 + ni) * 16 + E * kw + e) * 64 + lane;
          vv[q * (GATES ? NT : 1) + ni][e] = (red[idx] + red[idx + WSTRIDE]) + (red[idx + 2 * WSTRIDE] + red[idx + 3 * WSTRIDE]);
        }
    }
  }
  // ---- epilogue
  if (cell4) {
    if constexpr (CELL4) {
#pragma unroll
      for (int i = 0; i < IT4; ++i) {
        const int it = tid + 64 * NW * i;
        if (it >= MT * 256) break;
        const f32x4 v[4] = {vv[4 * i], vv[4 * i + 1], vv[4 * i + 2], vv[4 * i + 3]};
        g.ep.cell4(m0 + (it >> 3), n0 + 4 * (it & 7), v, pre4[i]);
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int mt = GATES ? hf * NP + q : (hf * TPW + q) / NT, ni0 = GATES ? 0 : (hf * TPW + q) % NT;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float v[GATES ? NT : 1];
#pragma unroll
      for (int ni = 0; ni < (GATES ? NT : 1); ++ni) v[ni] = vv[q * (GATES ? NT : 1) + ni][e];
      g.ep.template elem<(GATES ? NT : 1)>(m0 + 32 * mt + erow + e, n0 + 32 * ni0 + r, 32, v, pre[q][e]);
    }
  }
}

}  // namespace aocr
