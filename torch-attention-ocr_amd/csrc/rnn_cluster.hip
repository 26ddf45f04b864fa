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
__device__ __forceinline__ unsigned quad_swap(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }   // lane ^ 1
}  // namespace

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int G, int RT, bool CTX>
__global__ __launch_bounds__(256, 1) void enc_cl_fwd_kernel(EncClFwdArgs p) {
  constexpr int He = 64 * G, KS = 2 * G, R = 16 * RT;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % G, gl = (i8 / G) * 8 + xcd;           // the G members of a group: blocks 8 apart = one XCD (speed only)
  if (gl >= p.ngid) return;
  const int gid = p.gid0 + gl;
  const int dir = gid / p.groups, group = gid - dir * p.groups;
  const EncSeqDir& d = p.d[dir];
  const int B = p.B, T = p.T, row0 = group * R;
  const int ucol = 64 * member + 16 * wave + c16;               // this lane's hidden unit (all four gates)

  bf16x8 wres[4][KS];                                            // resident B fragments: gate g, k-step s
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int s = 0; s < KS; ++s) wres[g][s] = *reinterpret_cast<const bf16x8*>(d.w + (size_t)(g * He + ucol) * He + 32 * s + 8 * q);

  float cst[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) cst[rt][i] = 0.f;
  u64* const xg = p.xbuf + (size_t)gid * 2 * G * R * 32;         // [parity][member][row][32 granules]

  float zx[RT][4][4];
  auto load_zx = [&](int t) {
    const float* z = d.zx + (size_t)t * B * 4 * He + ucol;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(row0 + 16 * rt + 4 * q + i, B - 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) zx[rt][g][i] = z[(size_t)row * 4 * He + g * He];
      }
  };
  load_zx(d.reverse ? T - 1 : 0);
  bool dead = false;

  for (int it = 0; it < T && !dead; ++it) {
    const int t = d.reverse ? T - 1 - it : it;
    f32x4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[rt][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (it > 0) {
      const unsigned tag = p.epoch * 4096u + (unsigned)it;
      const u64* xp = xg + (size_t)((it - 1) & 1) * G * R * 32;
#pragma unroll
      for (int m = 0; m < G; ++m) {                              // h(t-1) slice of member m: units 64m .. 64m+63 = k-steps 2m, 2m+1
        const u64* xm = xp + (size_t)m * R * 32 + c16 * 32 + 4 * q;
        u64 gr[RT][2][4];
        int spins = 0;
        while (true) {
          bool ok = true;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int j = 0; j < 4; ++j) gr[rt][ks][j] = ld_granule(xm + rt * 16 * 32 + ks * 16 + j);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int j = 0; j < 4; ++j) ok = ok && (unsigned)(gr[rt][ks][j] >> 32) == tag;
          if (__all(ok)) break;
          if (++spins > CL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(p.err, 1); break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (dead) break;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 raw = {(unsigned)gr[rt][ks][0], (unsigned)gr[rt][ks][1], (unsigned)gr[rt][ks][2], (unsigned)gr[rt][ks][3]};
            bf16x8 a; __builtin_memcpy(&a, &raw, 16);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[rt][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wres[g][2 * m + ks], acc[rt][g], 0, 0, 0);
          }
      }
      if (dead) break;
    }
    // ---- gate math; the exchange stores go first (they are the group's critical path)
    const unsigned tagn = p.epoch * 4096u + (unsigned)(it + 1);
    u64* const xw = xg + ((size_t)(it & 1) * G + member) * R * 32 + (16 * wave + c16) / 2;
    float ig[RT][4], fg[RT][4], og[RT][4], gg[RT][4], hh[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ig[rt][i] = sigmoidf_(acc[rt][0][i] + zx[rt][0][i]); fg[rt][i] = sigmoidf_(acc[rt][1][i] + zx[rt][1][i]);
        og[rt][i] = sigmoidf_(acc[rt][2][i] + zx[rt][2][i]); gg[rt][i] = tanhf_(acc[rt][3][i] + zx[rt][3][i]);
        const float cn = fg[rt][i] * cst[rt][i] + ig[rt][i] * gg[rt][i];
        cst[rt][i] = cn; hh[rt][i] = og[rt][i] * tanhf_(cn);
        const unsigned hb = bf16_bits(hh[rt][i]), ot = quad_swap(hb);
        if (!(c16 & 1) && it + 1 < T) st_granule(xw + (size_t)(16 * rt + 4 * q + i) * 32, ((u64)tagn << 32) | (u64)(hb | (ot << 16)));
      }
    // ---- what the rest of the step needs in HBM: state slots, saved gates, bf16 shadow, context slice (EpGatesFwd's outputs)
    const size_t so = (size_t)(t + 1) * B * He;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 16 * rt + 4 * q + i;
        if (row < B) {
          const size_t o = so + (size_t)row * He + ucol;
          d.cs[o] = cst[rt][i]; d.hs[o] = hh[rt][i]; d.hsb[o] = (bf16_t)hh[rt][i];
          if (CTX) d.ctx[((size_t)row * T + t) * p.Hd + ucol] = hh[rt][i];
          float* gp = d.gates + ((size_t)t * B + row) * 4 * He + ucol;
          gp[0] = ig[rt][i]; gp[He] = fg[rt][i]; gp[2 * He] = og[rt][i]; gp[3 * He] = gg[rt][i];
        }
      }
    if (it + 1 < T) load_zx(d.reverse ? t - 1 : t + 1);
  }
}

// ---------------------------------------------------------------------------------------------
// backward (BPTT) of one layer
// ---------------------------------------------------------------------------------------------
template <int G, int RT>
__global__ __launch_bounds__(256, 1) void enc_cl_bwd_kernel(EncClBwdArgs p) {
  constexpr int He = 64 * G, KG = 4 * He, R = 16 * RT, AP = 256 * 2 + 16;     // A operand: [R][256 own gate columns] bf16, padded pitch
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const abuf = lds;                                            // [2][R][AP]
  f32x4* const own = reinterpret_cast<f32x4*>(lds + 2 * R * AP);              // [4 tiles][RT][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, q = lane >> 4;
  const int wid = blockIdx.x, xcd = wid & 7, i8 = wid >> 3;
  const int member = i8 % G, gl = (i8 / G) * 8 + xcd;
  if (gl >= p.ngid) return;
  const int gid = p.gid0 + gl;
  const int dir = gid / p.groups, group = gid - dir * p.groups;
  const EncSeqBwdDir& d = p.d[dir];
  const int B = p.B, T = p.T, row0 = group * R;
  const int ucol = 64 * member + 16 * wave + c16;               // epilogue: this lane's hidden unit

  // resident B fragments: this wave's G output tiles (units 16 (wave G + j) ..), K = the workgroup's own 256 gate columns
  // k = gate * 64 + local unit  <->  column gate * He + 64 member + local unit of W^T [He][4He]
  bf16x8 wres[G][8];
#pragma unroll
  for (int j = 0; j < G; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      wres[j][s] = *reinterpret_cast<const bf16x8*>(d.wt + (size_t)(16 * (wave * G + j) + c16) * KG + (s >> 1) * He + 64 * member + 32 * (s & 1) + 8 * q);

  float dcr[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = min(row0 + 16 * rt + 4 * q + i, B - 1); dcr[rt][i] = d.dc[(size_t)row * He + ucol]; }

  float pg[RT][4][4], pcc[RT][4], pcp[RT][4], pdh[RT][4];        // the next step's epilogue inputs
  auto prefetch = [&](int it) {
    const int t = d.forward_dir ? T - 1 - it : it, prev = d.forward_dir ? t : t + 2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(row0 + 16 * rt + 4 * q + i, B - 1);
        const float* g_t = d.gates + ((size_t)t * B + row) * KG + ucol;
#pragma unroll
        for (int g = 0; g < 4; ++g) pg[rt][g][i] = g_t[g * He];
        pcc[rt][i] = d.cs[((size_t)(t + 1) * B + row) * He + ucol];
        pcp[rt][i] = d.cs[((size_t)prev * B + row) * He + ucol];
        pdh[rt][i] = d.dh1[(size_t)row * d.dh1_row + (size_t)t * d.dh1_t + ucol];
      }
  };
  prefetch(0);
  if (d.dh2) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const int row = min(row0 + 16 * rt + 4 * q + i, B - 1); pdh[rt][i] += d.dh2[(size_t)row * d.dh2_row + ucol]; }
  }
  u64* const pb = p.pbuf + (size_t)gid * 2 * G * G * 4 * RT * 256;            // [parity][dest][src][tile][rt][lane][4 granules]
  bool dead = false;

  for (int it = 0; it < T; ++it) {
    const int t = d.forward_dir ? T - 1 - it : it;
    const unsigned tag = p.epoch * 4096u + (unsigned)it;
    const int par = it & 1;
    if (it > 0 && !dead) {
      // ---- partial d h for ALL units from this workgroup's own d z of the step before
      f32x4 acc[RT][G];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < G; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* ab = abuf + (size_t)((it - 1) & 1) * R * AP;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(ab + (size_t)(16 * rt + c16) * AP + (32 * s + 8 * q) * 2);
#pragma unroll
          for (int j = 0; j < G; ++j) acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wres[j][s], acc[rt][j], 0, 0, 0);
        }
      // ---- reduce-scatter: tile (wave G + j) belongs to member (wave G + j) / 4
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int nt = wave * G + j, dm = nt >> 2, e = nt & 3;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          if (dm == member) own[(e * RT + rt) * 64 + lane] = acc[rt][j];
          else {
            u64* dst = pb + ((((size_t)(par * G + dm) * G + member) * 4 + e) * RT + rt) * 256 + lane * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) st_granule(dst + i, ((u64)tag << 32) | (u64)__float_as_uint(acc[rt][j][i]));
          }
        }
      }
    }
    __syncthreads();                                             // own partials are in LDS; everyone is done with abuf[(it-1)&1]
    float dh[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) dh[rt][i] = pdh[rt][i];
    if (it > 0 && !dead) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const f32x4 o = own[(wave * RT + rt) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) dh[rt][i] += o[i];
      }
#pragma unroll
      for (int sm = 0; sm < G; ++sm) {
        if (sm == member) continue;
        const u64* src = pb + ((((size_t)(par * G + member) * G + sm) * 4 + wave) * RT) * 256 + lane * 4;
        u64 gr[RT][4];
        int spins = 0;
        while (true) {
          bool ok = true;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) gr[rt][i] = ld_granule(src + rt * 256 + i);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) ok = ok && (unsigned)(gr[rt][i] >> 32) == tag;
          if (__all(ok)) break;
          if (++spins > CL_SPIN_LIMIT) { dead = true; if (lane == 0) atomicExch(p.err, 2); break; }
          __builtin_amdgcn_s_sleep(1);
        }
        if (dead) break;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int i = 0; i < 4; ++i) dh[rt][i] += __uint_as_float((unsigned)gr[rt][i]);
      }
    }
    // ---- EpGatesBwd for this lane's cells: d z(t) to HBM (fp32 + bf16: operands of the hoisted gradients) and, as bf16, to LDS
    unsigned char* an = abuf + (size_t)par * R * AP;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rl = 16 * rt + 4 * q + i, row = row0 + rl;
        const float ig = pg[rt][0][i], fg = pg[rt][1][i], og = pg[rt][2][i], gg = pg[rt][3][i];
        const float tc = tanhf_(pcc[rt][i]);
        const float dc = dh[rt][i] * og * (1.f - tc * tc) + dcr[rt][i];
        const float d_o = dh[rt][i] * tc;
        const float z[4] = {dc * gg * ig * (1.f - ig), dc * pcp[rt][i] * fg * (1.f - fg), d_o * og * (1.f - og), dc * ig * (1.f - gg * gg)};
        dcr[rt][i] = dc * fg;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<bf16_t*>(an + (size_t)rl * AP + (g * 64 + 16 * wave + c16) * 2) = (bf16_t)z[g];
          if (row < B) {
            const size_t o = ((size_t)t * B + row) * KG + g * He + ucol;
            d.dz[o] = z[g]; d.dzb[o] = (bf16_t)z[g];
          }
        }
      }
    if (it + 1 < T) prefetch(it + 1);
    __syncthreads();                                             // d z(t) complete in LDS
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = row0 + 16 * rt + 4 * q + i; if (row < B) d.dc[(size_t)row * He + ucol] = dcr[rt][i]; }
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
  return true;
}
size_t enc_cluster_xbuf_bytes(int B, int He) {                   // forward exchange buffer for the largest plan (RT = 1 granularity covers RT = 2)
  const int G = He / 64, groups = (B + 15) / 16;
  return (size_t)2 * groups * 2 * G * 32 * 32 * sizeof(u64);     // [gid][parity][member][<= 32 rows][32 granules]
}
size_t enc_cluster_pbuf_bytes(int B, int He) {
  const int G = He / 64, groups = (B + 15) / 16;
  return (size_t)2 * groups * 2 * G * G * 4 * 2 * 256 * sizeof(u64);
}

template <int G, int RT> static void launch_fwd(hipStream_t s, const EncClFwdArgs& a, int grid) {
  if (a.d[0].ctx) hipLaunchKernelGGL((enc_cl_fwd_kernel<G, RT, true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((enc_cl_fwd_kernel<G, RT, false>), dim3(grid), dim3(256), 0, s, a);
}
template <int G, int RT> static void launch_bwd(hipStream_t s, const EncClBwdArgs& a, int grid) {
  const size_t lds = (size_t)2 * 16 * RT * (256 * 2 + 16) + (size_t)4 * RT * 64 * 16;
  hipLaunchKernelGGL((enc_cl_bwd_kernel<G, RT>), dim3(grid), dim3(256), lds, s, a);
}
static int cluster_cus() {
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return cus;
}
void enc_cluster_forward(hipStream_t s, const EncClFwdArgs& a0, int G, int RT) {
  const int per_pass = std::max(8, cluster_cus() / (8 * G) * 8);          // groups (gids) one launch can keep resident
  for (int g0 = 0; g0 < 2 * a0.groups; g0 += per_pass) {
    EncClFwdArgs a = a0; a.gid0 = g0; a.ngid = std::min(per_pass, 2 * a0.groups - g0);
    const int grid = 8 * G * ((a.ngid + 7) / 8);
#define AOCR_CL(GG) do { if (RT == 1) launch_fwd<GG, 1>(s, a, grid); else launch_fwd<GG, 2>(s, a, grid); } while (0)
    if (G == 1) AOCR_CL(1); else if (G == 2) AOCR_CL(2); else if (G == 4) AOCR_CL(4); else launch_fwd<8, 1>(s, a, grid);
#undef AOCR_CL
  }
}
void enc_cluster_backward(hipStream_t s, const EncClBwdArgs& a0, int G, int RT) {
  const int per_pass = std::max(8, cluster_cus() / (8 * G) * 8);
  for (int g0 = 0; g0 < 2 * a0.groups; g0 += per_pass) {
    EncClBwdArgs a = a0; a.gid0 = g0; a.ngid = std::min(per_pass, 2 * a0.groups - g0);
    const int grid = 8 * G * ((a.ngid + 7) / 8);
#define AOCR_CL(GG) do { if (RT == 1) launch_bwd<GG, 1>(s, a, grid); else launch_bwd<GG, 2>(s, a, grid); } while (0)
    if (G == 1) AOCR_CL(1); else if (G == 2) AOCR_CL(2); else if (G == 4) AOCR_CL(4); else launch_bwd<8, 1>(s, a, grid);
#undef AOCR_CL
  }
}

}  // namespace aocr
