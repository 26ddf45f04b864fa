// Experiment (round 3): 256 x 256 x 32 LDS-DMA GEMM with FOUR waves per workgroup, each 128 x 128 (16 accumulator tiles of 32 x 32 =
// 256 AGPRs), one wave per SIMD, against the 8-wave / 128 x 64 form of gemm_dma_bf16_kernel.  Per K step the 8-wave form reads
// 96 KB of fragments from LDS (A rows by 4 waves, B rows by 2) and writes 32 KB: 1024 LDS cycles at 128 B / clk -- as many as the
// step's 1024 MFMA cycles per SIMD; the 4-wave form reads 64 KB.  C[M][N] = A[M][K] . B[N][K]^T, bf16, K contiguous.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../torch-attention-ocr_amd/csrc gemm4w.hip -o gemm4w
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "ops.h"
#include "mfma_gemm.h"
using namespace aocr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ unsigned long long g_probe[2];
static int g_warm = 3;             // launches before every timed loop (argv[1]; ~5000 = 1.5 s of sustained load: steady-state clock)
__global__ void fill(bf16_t* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16_t)(((h & 0xffff) / 32768.0f) - 1.0f);
  }
}

// MODE 0: full; 1: no in-loop DMA; 2: no MFMA; 4: no fragment reads
template <int MODE, int HALFBAR>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm, float* __restrict__ C, int M, int N, int K, int gx, int gy, int lda, int kwrap) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 32768];
  const int nwg = gx * gy, orig = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
  const int bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
  const int m_blk = (bid / gx) * 256, n_blk = (bid % gx) * 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
  // staging: a piece = 16 rows x 64 B; lane -> row (lane >> 2), position lane & 3 holding chunk (lane & 3) ^ ((row >> 2) & 3)
  // wave w stages A pieces 4w .. 4w+3 and B pieces 4w .. 4w+3 of every tile
  const int prow = lane >> 2, pchunk = (lane & 3) ^ ((lane >> 4) & 3);
  const bf16_t* asrc[4]; const bf16_t* bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    asrc[j] = A + (size_t)(m_blk + (4 * wave + j) * 16 + prow) * lda + pchunk * 8;
    bsrc[j] = Bm + (size_t)(n_blk + (4 * wave + j) * 16 + prow) * K + pchunk * 8;
  }
  const int nk = K >> 5;
  auto issue = [&](int kt) {                              // tile kt -> slot kt & 3 (kt >= nk: re-reads the last tile, harmless)
    const int k = (kt < nk ? kt : nk - 1) << 5;
    unsigned char* base = lds + (kt & 3) * 32768 + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(asrc[j] + (k & (kwrap - 1)), base + j * 1024);
#pragma unroll
    for (int j = 0; j < 4; ++j) dma16(bsrc[j] + k, base + 16384 + j * 1024);
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int swz = (r >> 2) & 3;
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    aoff[s] = (wm * 128 + r) * 64 + (((2 * s + h) ^ swz) << 4);
    boff[s] = 16384 + (wn * 128 + r) * 64 + (((2 * s + h) ^ swz) << 4);
  }
  issue(0); issue(1); issue(2);
  const unsigned long long pc0 = __builtin_readcyclecounter(), pw0 = wall_clock64();
  if constexpr (HALFBAR == 0) {
    // one barrier per tile: wait(own pieces of tile kt) -> barrier -> reads of tile kt, DMA of tile kt+3, 32 MFMAs
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned char* L = lds + (kt & 3) * 32768;
      bf16x8 af[2][4], bf[2][4];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (!(MODE & 4)) { af[s2][i] = *reinterpret_cast<const bf16x8*>(L + aoff[s2] + i * 2048); bf[s2][i] = *reinterpret_cast<const bf16x8*>(L + boff[s2] + i * 2048); }
          else { af[s2][i] = bf16x8{}; bf[s2][i] = bf16x8{}; }
        }
      }
      if constexpr (!(MODE & 1)) issue(kt + 3);
      if constexpr (!(MODE & 2)) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][mi], bf[s2][ni], acc[mi][ni], 0, 0, 0);
      }
    }
  } else if constexpr (HALFBAR == 1) {
    // fragments of the first k half of tile kt+1 are read under the MFMAs of the second half of tile kt: the barrier sits in the middle of a tile
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a0[i] = *reinterpret_cast<const bf16x8*>(lds + aoff[0] + i * 2048); b0[i] = *reinterpret_cast<const bf16x8*>(lds + boff[0] + i * 2048); }
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* L = lds + (kt & 3) * 32768;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a1[i] = *reinterpret_cast<const bf16x8*>(L + aoff[1] + i * 2048); b1[i] = *reinterpret_cast<const bf16x8*>(L + boff[1] + i * 2048); }
      if constexpr (!(MODE & 2)) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[mi], b0[ni], acc[mi][ni], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // own pieces of tile kt+1 landed (tile kt+2 in flight)
      __builtin_amdgcn_s_barrier();                         // everyone's; and everyone has read the second half of tile kt-1... and issued the reads of this tile's second half
      if constexpr (!(MODE & 1)) issue(kt + 3);             // -> slot of tile kt-1
      const unsigned char* Ln = lds + ((kt + 1) & 3) * 32768;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a0[i] = *reinterpret_cast<const bf16x8*>(Ln + aoff[0] + i * 2048); b0[i] = *reinterpret_cast<const bf16x8*>(Ln + boff[0] + i * 2048); }
      if constexpr (!(MODE & 2)) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mi], b1[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  }
  if constexpr (HALFBAR == 2) {
    // hand-scheduled: asm fragment reads with counted lgkmcnt waits; the DMA pieces of tile kt+3 in two groups of four between MFMA groups
    const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#define DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(dst) : "v"(addr))
    auto rd = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[4], int slot, int half) {
      const unsigned pa = lbase + slot * 32768 + aoff[half], pb = lbase + slot * 32768 + boff[half];
      DSR(fa[0], pa, 0); DSR(fb[0], pb, 0); DSR(fa[1], pa, 2048); DSR(fb[1], pb, 2048);
      DSR(fa[2], pa, 4096); DSR(fb[2], pb, 4096); DSR(fa[3], pa, 6144); DSR(fb[3], pb, 6144);
    };
#undef DSR
    auto issue_half = [&](int kt, int g) {                // g = 0: the four A pieces, 1: the four B pieces of this wave
      const int k = (kt < nk ? kt : nk - 1) << 5;
      unsigned char* base = lds + (kt & 3) * 32768 + wave * 4096 + g * 16384;
#pragma unroll
      for (int j = 0; j < 4; ++j) dma16(g == 0 ? asrc[j] + (k & (kwrap - 1)) : bsrc[j] + k, base + j * 1024);
    };
    auto mm = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], int m_lo, int m_hi) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        if (mi >= m_lo && mi < m_hi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    };
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rd(a0, b0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      if constexpr (!(MODE & 4)) { if (!(MODE & 8) || kt == 0) rd(a1, b1, kt & 3, 1); }
      if constexpr (MODE & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 2)) mm(a0, b0, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 1)) issue_half(kt + 3, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 2)) mm(a0, b0, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 1)) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 4) && !(MODE & 8)) rd(a0, b0, (kt + 1) & 3, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 2)) mm(a1, b1, 0, 2);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 1)) issue_half(kt + 3, 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(MODE & 2)) mm(a1, b1, 2, 4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (HALFBAR == 3) {
    // as HALFBAR == 2, with the eight fragment reads and four DMA pieces of a half SPREAD between its sixteen MFMAs (one read per
    // two MFMAs, one piece per four), every position pinned by sched_barrier
    const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
    bf16x8 a0[4], b0[4], a1[4], b1[4];
#define DSR1(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define SB() __builtin_amdgcn_sched_barrier(0)
    auto half = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4], bf16x8 (&na)[4], bf16x8 (&nb)[4], int rslot, int rhalf, int dkt, int dg, bool do_read) {
      const unsigned pa = lbase + rslot * 32768 + aoff[rhalf], pb = lbase + rslot * 32768 + boff[rhalf];
      const int k = (dkt < nk ? dkt : nk - 1) << 5;
      unsigned char* dbase = lds + (dkt & 3) * 32768 + wave * 4096 + dg * 16384;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          if constexpr (!(MODE & 2)) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
          SB();
          if constexpr (!(MODE & 4)) {
            if (do_read) {
              if (ni == 0) { if (mi == 0) DSR1(na[0], pa, 0); else if (mi == 1) DSR1(na[1], pa, 2048); else if (mi == 2) DSR1(na[2], pa, 4096); else DSR1(na[3], pa, 6144); }
              if (ni == 2) { if (mi == 0) DSR1(nb[0], pb, 0); else if (mi == 1) DSR1(nb[1], pb, 2048); else if (mi == 2) DSR1(nb[2], pb, 4096); else DSR1(nb[3], pb, 6144); }
            }
          }
          if constexpr (!(MODE & 1)) { if (ni == 3 && (!(MODE & 16) || dg == 1 || mi < 2)) dma16(dg == 0 ? asrc[mi] + (k & (kwrap - 1)) : bsrc[mi] + k, dbase + mi * 1024); }
          SB();
        }
      }
    };
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    DSR1(a0[0], lbase + aoff[0], 0); DSR1(a0[1], lbase + aoff[0], 2048); DSR1(a0[2], lbase + aoff[0], 4096); DSR1(a0[3], lbase + aoff[0], 6144);
    DSR1(b0[0], lbase + boff[0], 0); DSR1(b0[1], lbase + boff[0], 2048); DSR1(b0[2], lbase + boff[0], 4096); DSR1(b0[3], lbase + boff[0], 6144);
    for (int kt = 0; kt < nk; ++kt) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      SB();
      half(a0, b0, a1, b1, kt & 3, 1, kt + 3, 0, true);
      if constexpr (MODE & 16) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else
      if constexpr (!(MODE & 1)) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      SB();
      half(a1, b1, a0, b0, (kt + 1) & 3, 0, kt + 3, 1, true);
    }
#undef DSR1
#undef SB
  }

  if constexpr (HALFBAR == 4) {
    // Round 4: the same 4-wave / 128 x 128 tile on v_mfma_f32_16x16x32_bf16 -- 8 x 8 accumulator tiles of 16 x 16 (256 AGPRs as before), an A / B fragment
    // = 16 rows x the WHOLE 32-deep k tile (lane l: row l % 16, k-chunk l / 16), 64 MFMAs of 16 cycles per tile = the same 1024 MFMA cycles and the same
    // 16 ds_read_b128 per wave and tile.  /opt/skills/guides/MI355X_MICROARCH.md (DVFS give-back, item 7): under the power cap the 16x16x32 loop holds a
    // clock ~1.12-1.15 x that of the 32x32x16 loop.  ONE barrier per tile: block kt multiplies F(kt), reads F(kt+1) (slot (kt+1) & 3) and issues the DMA
    // of tile kt+4 into slot kt & 3 (read during block kt-1); every read (one per four MFMAs) and DMA piece (one per eight) pinned between two MFMAs.
    // LDS image unchanged (piece position p of row r holds k-chunk p ^ ((r >> 2) & 3)); the 16 x 4-chunk fragment read is conflict-free on it when
    // fragment row i sits on physical row 4 sigma(i >> 2) + (i & 3), sigma = (0, 3, 2, 1) (tools/bank16.py); the output rows / columns permute with it.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const unsigned lbase = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)lds);
    const int i16 = lane & 15, g16 = lane >> 4;
    const int sg = (i16 >> 2) == 1 ? 3 : ((i16 >> 2) == 3 ? 1 : (i16 >> 2));
    const int prow16 = sg * 4 + (i16 & 3);
    const unsigned fa0 = lbase + (wm * 128 + prow16) * 64 + ((g16 ^ sg) << 4), fb0 = lbase + 16384 + (wn * 128 + prow16) * 64 + ((g16 ^ sg) << 4);
    f32x4 c4[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) c4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 pa[8], pb[8], qa[8], qb[8];
#define DSR4(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define SB4() __builtin_amdgcn_sched_barrier(0)
    auto rd1 = [&](bf16x8 (&na)[8], bf16x8 (&nb)[8], unsigned pa_, unsigned pb_, int q) {     // read q (0..15) of a tile's fragments: a0 b0 a1 b1 ...
      const int i = q >> 1;
      if (q & 1) { if (i == 0) DSR4(nb[0], pb_, 0); else if (i == 1) DSR4(nb[1], pb_, 1024); else if (i == 2) DSR4(nb[2], pb_, 2048); else if (i == 3) DSR4(nb[3], pb_, 3072);
                   else if (i == 4) DSR4(nb[4], pb_, 4096); else if (i == 5) DSR4(nb[5], pb_, 5120); else if (i == 6) DSR4(nb[6], pb_, 6144); else DSR4(nb[7], pb_, 7168); }
      else { if (i == 0) DSR4(na[0], pa_, 0); else if (i == 1) DSR4(na[1], pa_, 1024); else if (i == 2) DSR4(na[2], pa_, 2048); else if (i == 3) DSR4(na[3], pa_, 3072);
             else if (i == 4) DSR4(na[4], pa_, 4096); else if (i == 5) DSR4(na[5], pa_, 5120); else if (i == 6) DSR4(na[6], pa_, 6144); else DSR4(na[7], pa_, 7168); }
    };
    auto block = [&](const bf16x8 (&ca)[8], const bf16x8 (&cb)[8], bf16x8 (&na)[8], bf16x8 (&nb)[8], int kt) {
      const unsigned so = (unsigned)(((kt + 1) & 3) * 32768);
      const unsigned ra = fa0 + so, rb = fb0 + so;
      const int dkt = kt + 4, k = (dkt < nk ? dkt : nk - 1) << 5;
      unsigned char* dbase = lds + (dkt & 3) * 32768 + wave * 4096;
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          if constexpr (!(MODE & 2)) c4[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[mi], cb[ni], c4[mi][ni], 0, 0, 0);
          SB4();
          const int q = mi * 8 + ni;
          if constexpr (!(MODE & 4)) { if ((q & 3) == 1 && (!(MODE & 8) || kt < 0)) rd1(na, nb, ra, rb, q >> 2); }
          if constexpr (!(MODE & 1)) { if ((q & 7) == 6 && (!(MODE & 16) || (q >> 3) >= 2)) { const int j = (q >> 3) & 3; dma16((q >> 3) < 4 ? asrc[j] + (k & (kwrap - 1)) : bsrc[j] + k, dbase + ((q >> 3) < 4 ? 0 : 16384) + j * 1024); } }
          SB4();
        }
      }
    };
    issue(3);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) rd1(pa, pb, fa0, fb0, q);
    for (int kt = 0; kt < nk; kt += 2) {
      if constexpr (MODE & 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else if constexpr (MODE & 16) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      SB4();
      block(pa, pb, qa, qb, kt);
      if constexpr (MODE & 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else if constexpr (MODE & 16) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      SB4();
      block(qa, qb, pa, pb, kt + 1);
    }
#undef DSR4
#undef SB4
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(pa[i]), "v"(pb[i]), "v"(qa[i]), "v"(qb[i]));
    if (blockIdx.x == 17 && tid == 0) { g_probe[0] = __builtin_readcyclecounter() - pc0; g_probe[1] = wall_clock64() - pw0; }
    // D tile (mi, ni): lane l holds fragment column j = l % 16 (B row) and fragment rows 4 (l / 16) + v (A rows), v = 0..3
    const int sgr = g16 == 1 ? 3 : (g16 == 3 ? 1 : g16);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int row = m_blk + wm * 128 + 16 * mi + 4 * sgr + v, col = n_blk + wn * 128 + 16 * ni + prow16;
          if (row < M && col < N) C[(size_t)row * N + col] = c4[mi][ni][v];
        }
    return;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (blockIdx.x == 17 && tid == 0) { g_probe[0] = __builtin_readcyclecounter() - pc0; g_probe[1] = wall_clock64() - pw0; }
  const int m0 = m_blk + wm * 128, n0 = n_blk + wn * 128;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = m0 + 32 * mi + 8 * q + 4 * h + i, col = n0 + 32 * ni + r;
          if (row < M && col < N) C[(size_t)row * N + col] = acc[mi][ni][4 * q + i];
        }
}

template <int MODE, int HALFBAR> static int run(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K, const char* name, int lda = 4608, int kwrap = 8192) {
  const int gx = N / 256, gy = M / 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < g_warm; ++i) hipLaunchKernelGGL((gemm4w_kernel<MODE, HALFBAR>), dim3(gx * gy), dim3(256), 0, 0, A, B, C, M, N, K, gx, gy, lda, kwrap);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gemm4w_kernel<MODE, HALFBAR>), dim3(gx * gy), dim3(256), 0, 0, A, B, C, M, N, K, gx, gy, lda, kwrap);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / it;
  unsigned long long pr[2]; CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), 16));
  printf("%-52s %8.1f us  %7.1f TFLOP/s   K loop of one workgroup: %.0f shader cycles in %.1f us = %.2f GHz\n", name, us, 2.0 * M * N * K / us / 1e6, (double)pr[0], pr[1] / 100.0, pr[0] / (pr[1] * 10.0));
  return 0;
}
template <int ABL> static int run8(const LoadKh& a, const LoadKh& b, const EpStore& ep, int M, int N, int K, const bf16_t* zero, const char* name) {
  const int gx = N / 256, gy = M / 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_dma_bf16_kernel<LoadKh, LoadKh, EpStore, ABL, false, false>), dim3(gx * gy), dim3(512), 0, 0, a, b, ep, K, gx, gy, zero);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int it = 20;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gemm_dma_bf16_kernel<LoadKh, LoadKh, EpStore, ABL, false, false>), dim3(gx * gy), dim3(512), 0, 0, a, b, ep, K, gx, gy, zero);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / it;
  printf("%-52s %8.1f us  %7.1f TFLOP/s\n", name, us, 2.0 * M * N * K / us / 1e6);
  return 0;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc > 1) g_warm = atoi(argv[1]);
  const bool quick = argc > 2;                          // argv[2]: only the spread-schedule comparison of the two MFMA shapes
  printf("warm-up launches before every timed loop: %d\n", g_warm);
  const int M = 65536, N = 512, K = 4608;
  bf16_t *A, *B, *zero; float* C;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&zero, 64)); CK(hipMemset(zero, 0, 64));
  hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u);
  hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, B, (size_t)N * K, 2u);
  CK(hipDeviceSynchronize());
  printf("plain GEMM M %d N %d K %d (the conv6 shape without the im2col gather)\n", M, N, K);
  LoadKh a; a.p = A; a.ld = K; a.rows = M; a.K = K;
  LoadKh b; b.p = B; b.ld = K; b.rows = N; b.K = K;
  EpStore ep = make_store(C, N, M, N, nullptr, nullptr, 0);
  if (quick) {
    for (int rep = 0; rep < 2; ++rep) {
      run<0, 3>(A, B, C, M, N, K, "32x32x16: 4 waves x 128x128, reads + DMA spread between MFMAs", 512, 512);
      run<0, 4>(A, B, C, M, N, K, "16x16x32: 4 waves x 128x128, reads + DMA spread between MFMAs", 512, 512);
      run<16, 3>(A, B, C, M, N, K, "  32x32x16, halo-like DMA volume", 512, 512);
      run<16, 4>(A, B, C, M, N, K, "  16x16x32, halo-like DMA volume", 512, 512);
      run<9, 2>(A, B, C, M, N, K, "  32x32x16: MFMA on resident random fragments + barrier", 512, 512);
      run<9, 4>(A, B, C, M, N, K, "  16x16x32: MFMA on resident random fragments + barrier", 512, 512);
    }
    return 0;
  }
  for (int rep = 0; rep < 2; ++rep) {
    run8<0>(a, b, ep, M, N, K, zero, "8 waves x 128x64 (gemm_dma_bf16_kernel, plain reads)");
    run<0, 0>(A, B, C, M, N, K, "4 waves x 128x128, barrier per tile");
    run<0, 1>(A, B, C, M, N, K, "4 waves x 128x128, barrier mid-tile (pipelined reads)");
    run<1, 0>(A, B, C, M, N, K, "  4w: no in-loop DMA");
    run<2, 0>(A, B, C, M, N, K, "  4w: no MFMA");
    run<4, 0>(A, B, C, M, N, K, "  4w: no fragment reads");
    run<5, 0>(A, B, C, M, N, K, "  4w: MFMA + barrier only");
  }
  printf("A re-read nine times from a 512-column window (67 MB: the conv6 input volume)\n");
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>(A, B, C, M, N, K, "4 waves x 128x128, barrier per tile", 512, 512);
    run<0, 1>(A, B, C, M, N, K, "4 waves x 128x128, barrier mid-tile", 512, 512);
    run<1, 0>(A, B, C, M, N, K, "  4w: no in-loop DMA", 512, 512);
    run<2, 0>(A, B, C, M, N, K, "  4w: no MFMA", 512, 512);
    run<4, 0>(A, B, C, M, N, K, "  4w: no fragment reads", 512, 512);
    run<6, 0>(A, B, C, M, N, K, "  4w: DMA + barrier only", 512, 512);
    run<0, 2>(A, B, C, M, N, K, "4 waves x 128x128, hand-scheduled", 512, 512);
    run<0, 3>(A, B, C, M, N, K, "4 waves x 128x128, reads + DMA spread between MFMAs", 512, 512);
    run<16, 3>(A, B, C, M, N, K, "  spread: 2 A pieces + 4 B pieces per wave and tile (halo-like volume; results wrong)", 512, 512);
    run<1, 3>(A, B, C, M, N, K, "  spread: no in-loop DMA", 512, 512);
    run<4, 3>(A, B, C, M, N, K, "  spread: no fragment reads", 512, 512);
    run<1, 2>(A, B, C, M, N, K, "  hand: no in-loop DMA", 512, 512);
    run<4, 2>(A, B, C, M, N, K, "  hand: no fragment reads", 512, 512);
    run<9, 2>(A, B, C, M, N, K, "  hand: MFMA on resident random fragments + barrier", 512, 512);
    run<8, 2>(A, B, C, M, N, K, "  hand: the same + DMA stream", 512, 512);
    run<0, 4>(A, B, C, M, N, K, "16x16x32: 4 waves x 128x128, reads + DMA spread between MFMAs", 512, 512);
    run<16, 4>(A, B, C, M, N, K, "  16x16x32: 2 A pieces + 4 B pieces per wave and tile (halo-like volume; results wrong)", 512, 512);
    run<1, 4>(A, B, C, M, N, K, "  16x16x32: no in-loop DMA", 512, 512);
    run<4, 4>(A, B, C, M, N, K, "  16x16x32: no fragment reads", 512, 512);
    run<9, 4>(A, B, C, M, N, K, "  16x16x32: MFMA on resident random fragments + barrier", 512, 512);
    run<8, 4>(A, B, C, M, N, K, "  16x16x32: the same + DMA stream", 512, 512);
  }
  // spot check of the 4-wave kernel against the 8-wave one
  std::vector<float> c4((size_t)256 * N), c8((size_t)256 * N);
  hipLaunchKernelGGL((gemm4w_kernel<0, 3>), dim3((N / 256) * (M / 256)), dim3(256), 0, 0, A, B, C, M, N, K, N / 256, M / 256, K, 8192);
  CK(hipMemcpy(c4.data(), C + (size_t)1000 * 256 / 256 * 256 * N, c4.size() * 4, hipMemcpyDeviceToHost));
  hipLaunchKernelGGL((gemm_dma_bf16_kernel<LoadKh, LoadKh, EpStore, 0, false, false>), dim3((N / 256) * (M / 256)), dim3(512), 0, 0, a, b, ep, K, N / 256, M / 256, zero);
  CK(hipMemcpy(c8.data(), C + (size_t)1000 * 256 / 256 * 256 * N, c8.size() * 4, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < c4.size(); ++i) { md = std::max(md, (double)fabsf(c4[i] - c8[i])); mx = std::max(mx, (double)fabsf(c8[i])); }
  printf("4-wave vs 8-wave: max abs diff %.3e (max |c| %.3e)\n", md, mx);
  hipLaunchKernelGGL((gemm4w_kernel<0, 4>), dim3((N / 256) * (M / 256)), dim3(256), 0, 0, A, B, C, M, N, K, N / 256, M / 256, K, 8192);
  CK(hipMemcpy(c4.data(), C + (size_t)1000 * 256 / 256 * 256 * N, c4.size() * 4, hipMemcpyDeviceToHost));
  md = 0;
  for (size_t i = 0; i < c4.size(); ++i) md = std::max(md, (double)fabsf(c4[i] - c8[i]));
  printf("16x16x32 4-wave vs 8-wave: max abs diff %.3e (max |c| %.3e; the 16x16x32 instruction sums a k tile in a different order)\n", md, mx);
  return 0;
}
