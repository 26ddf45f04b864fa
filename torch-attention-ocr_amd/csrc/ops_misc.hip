// ops_misc.hip -- the HBM/L2-bound kernels of the hot path: first conv layer (K=9, fused normalise+conv+ReLU+pool),
// BatchNorm, un-pooling, the attention score/softmax/context core, LogSoftMax+NLL, reductions, embedding,
// clipped SGD and the beam-search bookkeeping.  All wave-width constants are 64 (gfx950).
#include "ops.h"

namespace aocr {
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// =============================================================================================
// conv1: cnn.lua:9-15.  x (B,H,W) raw 0..255 -> y (B,H/2,W/2,64) = maxpool2x2(relu(conv3x3((x-128)/128)+b)).
// K = 9: VALU work (36 FMAs per window and channel) over an HBM-bound y write; x stays in L1/L2.
// =============================================================================================
// lane = output channel, wave = a strip of up to 32 pooling windows of one image row (the layout of conv1_bwd_kernel below):
// the normalised 4 x (2S+2) patch of the strip goes through a wave-private LDS block and is read back as broadcasts.
constexpr int C1S = 32;                                         // windows per strip
__device__ __forceinline__ void conv1_stage(const float* __restrict__ xb, int py, int px0, int H, int W, int lane,
                                            float (&sx)[4][2 * C1S + 4]) {
  // columns 2*px0-1 .. 2*px0+2*C1S of rows 2*py-1 .. 2*py+2; zero padding applies to the normalised map (cnn.lua:9-12)
  for (int e = lane; e < 4 * (2 * C1S + 2); e += 64) {
    const int i = e / (2 * C1S + 2), j = e - i * (2 * C1S + 2);
    const int yy = 2 * py - 1 + i, xx = 2 * px0 - 1 + j;
    const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    sx[i][j] = ok ? (xb[(int64_t)yy * W + xx] + (-128.0f)) * (1.0f / 128) : 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool ROUTE>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, bf16_t* __restrict__ yb,
                                                        int B, int H, int W, int Hp, int Wp, uint16_t* __restrict__ route) {
  __shared__ float sx[4][4][2 * C1S + 4];                       // [wave][row][col]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = w[lane * 9 + k];
  const float bb = bias[lane];
  const int spr = (Wp + C1S - 1) / C1S;
  const int64_t nstrips = (int64_t)B * Hp * spr;
  for (int64_t st = (int64_t)blockIdx.x * 4 + wave; st < nstrips; st += (int64_t)gridDim.x * 4) {
    const int sp = (int)(st % spr); const int64_t t = st / spr; const int py = (int)(t % Hp), b = (int)(t / Hp);
    const int px0 = sp * C1S, nw = min(C1S, Wp - px0);
    conv1_stage(x + (int64_t)b * H * W, py, px0, H, W, lane, sx[wave]);
    const int64_t o0 = (((int64_t)b * Hp + py) * Wp + px0) * 64 + lane;
#pragma unroll 1
    for (int w0 = 0; w0 < nw; w0 += 4) {
      float p[4][10];                                           // (the packed-math form of conv1_bwd_pk_kernel measured SLOWER here: 39.5 -> 44.1 us -- without the 36 accumulate FMAs the pair assembly outweighs the 18 FMAs saved)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; j += 2) {
          const float2 v = *reinterpret_cast<const float2*>(&sx[wave][i][2 * w0 + j]);
          p[i][j] = v.x; p[i][j + 1] = v.y;
        }
      unsigned code = 0;                                        // four 4-bit routes: 0 = below the ReLU floor, 1 + (2 dy + dx) = the window's first strict maximum
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float best = 0.f;                                       // relu floor
        unsigned bi = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            float s2 = bb;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) s2 = fmaf(wk[kh * 3 + kw], p[dy + kh][2 * u + dx + kw], s2);
            if (ROUTE) bi = s2 > best ? (unsigned)(1 + 2 * dy + dx) << (4 * u) : bi;
            best = fmaxf(best, s2);
          }
        if (ROUTE) code |= bi;
        if (w0 + u < nw) {
          if (y) y[o0 + (int64_t)(w0 + u) * 64] = best;
          if (yb) yb[o0 + (int64_t)(w0 + u) * 64] = (bf16_t)best;
        }
      }
      if (ROUTE) route[(st * (C1S / 4) + (w0 >> 2)) * 64 + lane] = (uint16_t)code;
    }
    __builtin_amdgcn_wave_barrier();                            // the next strip overwrites this wave's block
  }
}

// gradWeight/gradBias of conv1 (gradInput of the image is never used: model.lua:692 discards it).
// Re-computes the 4 conv values of each window to route d(pooled) through pool+ReLU.
// lane = output channel, wave = a strip of up to 32 pooling windows of one image row: the 4 x (2S+2) input patch of the strip is
// normalised once into a wave-private LDS row block and read back as broadcasts (every lane needs the same 16 values per window),
// d(pooled) is one coalesced 256-byte row per window.  Per window and lane: 36 FMAs to re-evaluate the four conv outputs, the
// arg-max/ReLU routing as four masked copies of g, 36 FMAs into the nine tap accumulators -- no dynamic register indexing.
__global__ __launch_bounds__(256) void conv1_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ dyp,
                                                        float* __restrict__ dw, float* __restrict__ db, int B, int H, int W,
                                                        int Hp, int Wp, float* __restrict__ partial) {
  __shared__ float sx[4][4][2 * C1S + 4];                       // [wave][row][col]
  __shared__ float swave[4][640];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = w[lane * 9 + k];
  const float bb = bias[lane];
  float acc[9], accb = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  const int spr = (Wp + C1S - 1) / C1S;                         // strips per image row
  const int64_t nstrips = (int64_t)B * Hp * spr;
  for (int64_t st = (int64_t)blockIdx.x * 4 + wave; st < nstrips; st += (int64_t)gridDim.x * 4) {
    const int sp = (int)(st % spr); const int64_t t = st / spr; const int py = (int)(t % Hp), b = (int)(t / Hp);
    const int px0 = sp * C1S, nw = min(C1S, Wp - px0);
    conv1_stage(x + (int64_t)b * H * W, py, px0, H, W, lane, sx[wave]);
    const float* gp = dyp + (((int64_t)b * Hp + py) * Wp + px0) * 64 + lane;
#pragma unroll 1
    for (int w0 = 0; w0 < nw; w0 += 4) {
      float g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = (w0 + u < nw) ? gp[(int64_t)(w0 + u) * 64] : 0.f;
      float p[4][10];                                           // columns 2*w0 .. 2*w0+9 of the staged block (broadcast reads)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; j += 2) {
          const float2 v = *reinterpret_cast<const float2*>(&sx[wave][i][2 * w0 + j]);
          p[i][j] = v.x; p[i][j + 1] = v.y;
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float sv[4];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            float s2 = bb;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
              for (int kw = 0; kw < 3; ++kw) s2 = fmaf(wk[kh * 3 + kw], p[dy + kh][2 * u + dx + kw], s2);
            sv[dy * 2 + dx] = s2;
          }
        // first strict maximum above the ReLU floor, in the order (0,0),(0,1),(1,0),(1,1) -- the forward's fmaxf chain
        float best = 0.f; int bi = -1;
#pragma unroll
        for (int q = 0; q < 4; ++q) if (sv[q] > best) { best = sv[q]; bi = q; }
        const float gg = g[u];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float gq = bi == q ? gg : 0.f;
          const int dy = q >> 1, dx = q & 1;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(gq, p[dy + kh][2 * u + dx + kw], acc[kh * 3 + kw]);
        }
        accb += bi >= 0 ? gg : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();                            // the next strip overwrites this wave's block
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) swave[wave][lane * 10 + k] = acc[k];
  swave[wave][lane * 10 + 9] = accb;
  __syncthreads();
  for (int i = threadIdx.x; i < 640; i += 256) {
    const float t = (swave[0][i] + swave[1][i]) + (swave[2][i] + swave[3][i]);
    int c = i / 10, k = i % 10;
    if (partial) partial[(int64_t)blockIdx.x * 640 + (k < 9 ? c * 9 + k : 576 + c)] = t;     // per-workgroup slab, summed by colsum
    else if (k < 9) atomicAdd(&dw[c * 9 + k], t); else atomicAdd(&db[c], t);
  }
}

// The same with PACKED fp32 math (v_pk_fma_f32: two FMAs per instruction): the two columns dx = 0, 1 of a pooling window are the two
// halves of a register pair -- even pairs come straight from the LDS block, the odd ones (kernel column 1) are assembled from their
// neighbours.  Per window and lane 18 + 18 packed FMAs instead of 36 + 36; every conv value still sums its nine taps in the forward
// pass's order, so the arg-max / ReLU routing is unchanged; the tap accumulators are kept per column parity and added at the end.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool ROUTE>
__global__ __launch_bounds__(256) void conv1_bwd_pk_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ dyp,
                                                           float* __restrict__ dw, float* __restrict__ db, int B, int H, int W,
                                                           int Hp, int Wp, float* __restrict__ partial,
                                                           const uint16_t* __restrict__ route) {
  __shared__ float sx[4][4][2 * C1S + 4];                       // [wave][row][col]
  __shared__ float swave[4][640];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x2 wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { const float v = w[lane * 9 + k]; wk[k] = f32x2{v, v}; }
  const float bb = bias[lane];
  f32x2 acc[9]; float accb = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = f32x2{0.f, 0.f};
  const int spr = (Wp + C1S - 1) / C1S;
  const int64_t nstrips = (int64_t)B * Hp * spr;
  for (int64_t st = (int64_t)blockIdx.x * 4 + wave; st < nstrips; st += (int64_t)gridDim.x * 4) {
    const int sp = (int)(st % spr); const int64_t t = st / spr; const int py = (int)(t % Hp), b = (int)(t / Hp);
    const int px0 = sp * C1S, nw = min(C1S, Wp - px0);
    conv1_stage(x + (int64_t)b * H * W, py, px0, H, W, lane, sx[wave]);
    const float* gp = dyp + (((int64_t)b * Hp + py) * Wp + px0) * 64 + lane;
#pragma unroll 1
    for (int w0 = 0; w0 < nw; w0 += 4) {
      float g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = (w0 + u < nw) ? gp[(int64_t)(w0 + u) * 64] : 0.f;
      unsigned code = 0;
      if (ROUTE) code = route[(st * (C1S / 4) + (w0 >> 2)) * 64 + lane];
      f32x2 pe[4][5], po[4][4];                                 // even pairs (columns 2k, 2k+1 of the 10-column slice), odd pairs (2k+1, 2k+2)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < 5; ++k) pe[i][k] = *reinterpret_cast<const f32x2*>(&sx[wave][i][2 * w0 + 2 * k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) po[i][k] = f32x2{pe[i][k][1], pe[i][k + 1][0]};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int bi = -1;
        if (ROUTE) bi = (int)((code >> (4 * u)) & 7) - 1;       // the forward pass's own decision (conv1_fwd_kernel<true>)
        else {
          f32x2 sv[2];                                          // sv[dy] = conv outputs (dx = 0, dx = 1) of window row dy
#pragma unroll
          for (int dy = 0; dy < 2; ++dy) {
            f32x2 s2 = f32x2{bb, bb};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              s2 = __builtin_elementwise_fma(wk[kh * 3 + 0], pe[dy + kh][u], s2);
              s2 = __builtin_elementwise_fma(wk[kh * 3 + 1], po[dy + kh][u], s2);
              s2 = __builtin_elementwise_fma(wk[kh * 3 + 2], pe[dy + kh][u + 1], s2);
            }
            sv[dy] = s2;
          }
          // first strict maximum above the ReLU floor, in the order (0,0),(0,1),(1,0),(1,1) -- the forward's fmaxf chain
          float best = 0.f;
          if (sv[0][0] > best) { best = sv[0][0]; bi = 0; }
          if (sv[0][1] > best) { best = sv[0][1]; bi = 1; }
          if (sv[1][0] > best) { best = sv[1][0]; bi = 2; }
          if (sv[1][1] > best) { best = sv[1][1]; bi = 3; }
        }
        const float gg = g[u];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const f32x2 gq = f32x2{bi == 2 * dy ? gg : 0.f, bi == 2 * dy + 1 ? gg : 0.f};
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            acc[kh * 3 + 0] = __builtin_elementwise_fma(gq, pe[dy + kh][u], acc[kh * 3 + 0]);
            acc[kh * 3 + 1] = __builtin_elementwise_fma(gq, po[dy + kh][u], acc[kh * 3 + 1]);
            acc[kh * 3 + 2] = __builtin_elementwise_fma(gq, pe[dy + kh][u + 1], acc[kh * 3 + 2]);
          }
        }
        accb += bi >= 0 ? gg : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();                            // the next strip overwrites this wave's block
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) swave[wave][lane * 10 + k] = acc[k][0] + acc[k][1];
  swave[wave][lane * 10 + 9] = accb;
  __syncthreads();
  for (int i = threadIdx.x; i < 640; i += 256) {
    const float t = (swave[0][i] + swave[1][i]) + (swave[2][i] + swave[3][i]);
    int c = i / 10, k = i % 10;
    if (partial) partial[(int64_t)blockIdx.x * 640 + (k < 9 ? c * 9 + k : 576 + c)] = t;
    else if (k < 9) atomicAdd(&dw[c * 9 + k], t); else atomicAdd(&db[c], t);
  }
}

size_t conv1_route_elems(int B, int H, int W) { return (size_t)B * (H / 2) * ((W / 2 + C1S - 1) / C1S) * (C1S / 4) * 64; }
void conv1_forward(hipStream_t s, const float* x, const float* w, const float* bias, float* y, int B, int H, int W, bf16_t* yb,
                   uint16_t* route) {
  int Hp = H / 2, Wp = W / 2;
  int64_t strips = (int64_t)B * Hp * ((Wp + C1S - 1) / C1S);
  int blocks = (int)std::min<int64_t>((strips + 3) / 4, 4096);
  if (route) hipLaunchKernelGGL(conv1_fwd_kernel<true>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, yb, B, H, W, Hp, Wp, route);
  else hipLaunchKernelGGL(conv1_fwd_kernel<false>, dim3(blocks), dim3(256), 0, s, x, w, bias, y, yb, B, H, W, Hp, Wp, route);
}
void conv1_backward(hipStream_t s, const float* x, const float* w, const float* bias, const float* dyp, float* dw, float* db,
                    int B, int H, int W, float* scratch, ColsumJobs* defer, const uint16_t* route) {
  int Hp = H / 2, Wp = W / 2;
  int64_t strips = (int64_t)B * Hp * ((Wp + C1S - 1) / C1S);
  // with a scratch slab (>= 4096*640 floats) every workgroup writes its partial sums and two column sums finish the job:
  // no contended global atomics, more workgroups
  int blocks = (int)std::min<int64_t>((strips + 3) / 4, scratch ? 2048 : 1024);
  if (getenv("AOCR_CONV1_SCALAR")) hipLaunchKernelGGL(conv1_bwd_kernel, dim3(blocks), dim3(256), 0, s, x, w, bias, dyp, dw, db, B, H, W, Hp, Wp, scratch);
  else if (route) hipLaunchKernelGGL(conv1_bwd_pk_kernel<true>, dim3(blocks), dim3(256), 0, s, x, w, bias, dyp, dw, db, B, H, W, Hp, Wp, scratch, route);
  else hipLaunchKernelGGL(conv1_bwd_pk_kernel<false>, dim3(blocks), dim3(256), 0, s, x, w, bias, dyp, dw, db, B, H, W, Hp, Wp, scratch, route);
  if (scratch && defer) { colsum_defer(*defer, scratch, 640, blocks, 576, dw); colsum_defer(*defer, scratch + 576, 640, blocks, 64, db); }
  else if (scratch) {
    colsum_accum(s, scratch, 640, blocks, 576, dw);
    colsum_accum(s, scratch + 576, 640, blocks, 64, db);
  }
}

// =============================================================================================
// bf16 shadows of a conv weight tensor w [Cout][KK][Cin]: wb same layout (forward B operand, K-contiguous) and
// wtb [Cin][KK][Cout] (data-gradient B operand, K-contiguous over (tap, co)).  Refreshed once per step.
// =============================================================================================
__global__ __launch_bounds__(256) void conv_weight_shadow_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb,
                                                                 bf16_t* __restrict__ wtb, int Cout, int KK, int Cin) {
  const int64_t n = (int64_t)Cout * KK * Cin;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int ci = (int)(i % Cin); int64_t t = i / Cin; int tap = (int)(t % KK); int co = (int)(t / KK);
    bf16_t v = (bf16_t)w[i];
    wb[i] = v;
    wtb[((int64_t)ci * KK + tap) * Cout + co] = v;
  }
}
void conv_weight_shadows(hipStream_t s, const float* w, bf16_t* wb, bf16_t* wtb, int Cout, int KK, int Cin) {
  int64_t n = (int64_t)Cout * KK * Cin;
  hipLaunchKernelGGL(conv_weight_shadow_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, w, wb, wtb,
                     Cout, KK, Cin);
}

__global__ __launch_bounds__(256) void weight_shadow_kernel(const float* __restrict__ w, int64_t ld, int R, int C,
                                                            bf16_t* __restrict__ wb, bf16_t* __restrict__ wtb) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads
  for (int i = ty; i < 32; i += 8) {
    int r = r0 + i, c = c0 + tx;
    float v = (r < R && c < C) ? w[(int64_t)r * ld + c] : 0.f;
    tile[i][tx] = v;
    if (r < R && c < C) wb[(int64_t)r * C + c] = (bf16_t)v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) wtb[(int64_t)c * R + r] = (bf16_t)tile[tx][i];
  }
}
// All bf16 weight shadows of a step in ONE launch: a device-resident job table (built once: the pointers never change) lists
// 2-D pieces w [R][C] (leading dimension ld) -> wb [R][C] (ldb) and its transpose wtb [C][R] (ldt); a conv weight [Cout][tap][Cin]
// is one piece per tap (wtb is [Cin][tap][Cout]).  16 separate launches cost ~100 us of dispatch latency for ~20 us of traffic.
__global__ __launch_bounds__(256) void shadow_jobs_kernel(const ShadowJob* __restrict__ jobs, int njobs, int tile_off) {
  __shared__ float tile[32][33];
  const int bid = (int)blockIdx.x + tile_off;              // (a step launches the table in two parts: the pieces conv2 needs first, on an event of their own)
  int j = 0, hi = njobs;                                // uniform binary search: last job whose first tile is <= this workgroup's (tile0 ascends)
  while (hi - j > 1) { const int mid = (j + hi) >> 1; if (bid >= jobs[mid].tile0) j = mid; else hi = mid; }
  const ShadowJob J = jobs[j];
  const int t = bid - J.tile0;
  const int c0 = (t % J.tx) * 32, r0 = (t / J.tx) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  // 16-byte loads / 8-byte stores when the piece allows it (every matrix of the model does): one load and two stores per thread
  const bool vec = ((J.C | J.R | (int)J.ld | (int)J.ldb | (int)J.ldt) & 3) == 0 && ((reinterpret_cast<uintptr_t>(J.w) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(J.wb) & 7) == 0) && ((reinterpret_cast<uintptr_t>(J.wtb) & 7) == 0);
  if (vec) {
    const int q = threadIdx.x & 7, l = threadIdx.x >> 3;           // load: row l, columns 4q..4q+3; transposed store: column l, rows 4q..4q+3
    {
      const int r = r0 + l, c = c0 + 4 * q;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < J.R && c < J.C) {
        v = *reinterpret_cast<const float4*>(J.w + (int64_t)r * J.ld + c);
        bf16x4 hb; hb[0] = (bf16_t)v.x; hb[1] = (bf16_t)v.y; hb[2] = (bf16_t)v.z; hb[3] = (bf16_t)v.w;
        *reinterpret_cast<bf16x4*>(J.wb + (int64_t)r * J.ldb + c) = hb;
      }
      tile[l][4 * q] = v.x; tile[l][4 * q + 1] = v.y; tile[l][4 * q + 2] = v.z; tile[l][4 * q + 3] = v.w;
    }
    __syncthreads();
    {
      const int c = c0 + l, r = r0 + 4 * q;
      if (c < J.C && r < J.R) {
        bf16x4 hb; hb[0] = (bf16_t)tile[4 * q][l]; hb[1] = (bf16_t)tile[4 * q + 1][l]; hb[2] = (bf16_t)tile[4 * q + 2][l]; hb[3] = (bf16_t)tile[4 * q + 3][l];
        *reinterpret_cast<bf16x4*>(J.wtb + (int64_t)c * J.ldt + r) = hb;
      }
    }
    return;
  }
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    const float v = (r < J.R && c < J.C) ? J.w[(int64_t)r * J.ld + c] : 0.f;
    tile[i][tx] = v;
    if (r < J.R && c < J.C) J.wb[(int64_t)r * J.ldb + c] = (bf16_t)v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < J.R && c < J.C) J.wtb[(int64_t)c * J.ldt + r] = (bf16_t)tile[tx][i];
  }
}
void shadow_jobs(hipStream_t s, const ShadowJob* jobs_dev, int njobs, int total_tiles, int tile_off) {
  if (njobs > 0 && total_tiles > 0) hipLaunchKernelGGL(shadow_jobs_kernel, dim3(total_tiles), dim3(256), 0, s, jobs_dev, njobs, tile_off);
}
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ w, int64_t ld, int R, int C, float* __restrict__ wt) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) { int r = r0 + i, c = c0 + tx; tile[i][tx] = (r < R && c < C) ? w[(int64_t)r * ld + c] : 0.f; }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) { int c = c0 + i, r = r0 + tx; if (r < R && c < C) wt[(int64_t)c * R + r] = tile[tx][i]; }
}
void transpose_f32(hipStream_t s, const float* w, int64_t ld, int R, int C, float* wt) {
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, s, w, ld, R, C, wt);
}
__global__ __launch_bounds__(256) void conv_weight_transpose_f32_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int KK, int Cin) {
  const int64_t n = (int64_t)Cout * KK * Cin;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    int ci = (int)(i % Cin); int64_t t = i / Cin; int tap = (int)(t % KK); int co = (int)(t / KK);
    wt[((int64_t)ci * KK + tap) * Cout + co] = w[i];
  }
}
void conv_weight_transpose_f32(hipStream_t s, const float* w, float* wt, int Cout, int KK, int Cin) {
  int64_t n = (int64_t)Cout * KK * Cin;
  hipLaunchKernelGGL(conv_weight_transpose_f32_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, w, wt, Cout, KK, Cin);
}
void weight_shadows(hipStream_t s, const float* w, int64_t ld, int R, int C, bf16_t* wb, bf16_t* wtb) {
  hipLaunchKernelGGL(weight_shadow_kernel, dim3(cdiv(C, 32), cdiv(R, 32)), dim3(256), 0, s, w, ld, R, C, wb, wtb);
}

// =============================================================================================
// un-pool + ReLU backward: dy (B,Ho,Wo,C) from d(pooled), arg-max index and the pooled value (>0 <=> ReLU passed).
// =============================================================================================
// dy == nullptr (bf16 mode): only the bf16 shadow is written -- every consumer of the un-pooled gradient (data gradient, filter
// gradient) reads the shadow -- and the bias gradient sum_pixels dy[.,c] = sum_windows g[.,c]*(pooled>0) is accumulated here
// (dbias != nullptr) instead of by a column-sum pass over a 4x larger fp32 tensor.
template <bool F32OUT, bool BIAS>
__global__ __launch_bounds__(256) void unpool_kernel(const float* __restrict__ dp, const float* __restrict__ pooled,
                                                     const uint8_t* __restrict__ idx, float* __restrict__ dy, bf16_t* __restrict__ dyb,
                                                     float* __restrict__ dbias, int B, int Ho, int Wo, int C, int pool, int Hp, int Wp,
                                                     const bf16_t* __restrict__ pooledb) {
  const int C4 = C >> 2;
  const int64_t total = (int64_t)B * Hp * Wp * C4;
  float bs[4] = {0.f, 0.f, 0.f, 0.f};                 // BIAS: the grid stride is a multiple of C4, so a thread keeps its channel quad
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c4 = (int)(id % C4); int64_t win = id / C4;
    int px = (int)(win % Wp); int64_t t = win / Wp; int py = (int)(t % Hp); int b = (int)(t / Hp);
    float4 g = *reinterpret_cast<const float4*>(dp + win * C + c4 * 4);
    float4 pv;                                            // only its sign is used (ReLU mask): the bf16 shadow of the pooled map has it
    if (pooledb) { bf16x4 t4 = *reinterpret_cast<const bf16x4*>(pooledb + win * C + c4 * 4); pv = make_float4((float)t4[0], (float)t4[1], (float)t4[2], (float)t4[3]); }
    else pv = *reinterpret_cast<const float4*>(pooled + win * C + c4 * 4);
    uint32_t ii = *reinterpret_cast<const uint32_t*>(idx + win * C + c4 * 4);
    float gv[4] = {pv.x > 0.f ? g.x : 0.f, pv.y > 0.f ? g.y : 0.f, pv.z > 0.f ? g.z : 0.f, pv.w > 0.f ? g.w : 0.f};
    int iv[4] = {(int)(ii & 255), (int)((ii >> 8) & 255), (int)((ii >> 16) & 255), (int)(ii >> 24)};
    if (BIAS) { bs[0] += gv[0]; bs[1] += gv[1]; bs[2] += gv[2]; bs[3] += gv[3]; }
    const int npos = pool == 1 ? 4 : 2;
    for (int pos = 0; pos < npos; ++pos) {
      int y = 2 * py + (pool == 1 ? (pos >> 1) : pos);
      int x = pool == 1 ? 2 * px + (pos & 1) : px;
      float4 o = make_float4(iv[0] == pos ? gv[0] : 0.f, iv[1] == pos ? gv[1] : 0.f, iv[2] == pos ? gv[2] : 0.f,
                             iv[3] == pos ? gv[3] : 0.f);
      const int64_t off = (((int64_t)b * Ho + y) * Wo + x) * C + c4 * 4;
      if (F32OUT) *reinterpret_cast<float4*>(dy + off) = o;
      if (dyb) { bf16x4 hb; hb[0] = (bf16_t)o.x; hb[1] = (bf16_t)o.y; hb[2] = (bf16_t)o.z; hb[3] = (bf16_t)o.w; *reinterpret_cast<bf16x4*>(dyb + off) = hb; }
    }
  }
  if (BIAS) {                                           // threads t, t + C4, ... of the workgroup hold the same channel quad (256 % C4 == 0)
    __shared__ float sh[256][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = bs[k];
    __syncthreads();
    if ((int)threadIdx.x < C4) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int t = threadIdx.x; t < 256; t += C4)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += sh[t][k];
      // one partial row per workgroup (2048 same-address atomics per channel serialise: measured 216 us instead of ~60)
      *reinterpret_cast<float4*>(dbias + (size_t)blockIdx.x * C + threadIdx.x * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}
// the bf16-mode form (shadow only + bias partials) with EIGHT channels per thread: 16-byte bf16 stores and loads, two 16-byte gradient loads
// (the 4-channel form moves 8-byte pieces: 3.4 TB/s on the 126 MB of conv6's pass)
__global__ __launch_bounds__(256) void unpool8_kernel(const float* __restrict__ dp, const uint8_t* __restrict__ idx, bf16_t* __restrict__ dyb, float* __restrict__ dbias,
                                                      int B, int Ho, int Wo, int C, int pool, int Hp, int Wp, const bf16_t* __restrict__ pooledb, const bf16_t* __restrict__ dph = nullptr /* d(pooled) as bf16 */) {
  const int C8 = C >> 3;
  const int64_t total = (int64_t)B * Hp * Wp * C8;
  float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // the grid stride is a multiple of C8: a thread keeps its eight channels
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(id % C8); const int64_t win = id / C8;
    const int px = (int)(win % Wp); const int64_t t = win / Wp; const int py = (int)(t % Hp), b = (int)(t / Hp);
    const int64_t wo = win * C + c8 * 8;
    float g[8];
    if (dph) { const bf16x8 gh = *reinterpret_cast<const bf16x8*>(dph + wo);
#pragma unroll
      for (int k = 0; k < 8; ++k) g[k] = (float)gh[k]; }
    else { const float4 g0 = *reinterpret_cast<const float4*>(dp + wo), g1 = *reinterpret_cast<const float4*>(dp + wo + 4);
      g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w; }
    const bf16x8 pv = *reinterpret_cast<const bf16x8*>(pooledb + wo);
    const uint2 ii = *reinterpret_cast<const uint2*>(idx + wo);
    float gv[8]; int iv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      gv[k] = (float)pv[k] > 0.f ? g[k] : 0.f;
      iv[k] = (int)(((k < 4 ? ii.x : ii.y) >> (8 * (k & 3))) & 255u);
      bs[k] += gv[k];
    }
    const int npos = pool == 1 ? 4 : 2;
    for (int pos = 0; pos < npos; ++pos) {
      const int y = 2 * py + (pool == 1 ? (pos >> 1) : pos), x = pool == 1 ? 2 * px + (pos & 1) : px;
      bf16x8 hb;
#pragma unroll
      for (int k = 0; k < 8; ++k) hb[k] = (bf16_t)(iv[k] == pos ? gv[k] : 0.f);
      *reinterpret_cast<bf16x8*>(dyb + (((int64_t)b * Ho + y) * Wo + x) * C + c8 * 8) = hb;
    }
  }
  __shared__ float sh[256][9];
#pragma unroll
  for (int k = 0; k < 8; ++k) sh[threadIdx.x][k] = bs[k];
  __syncthreads();
  if ((int)threadIdx.x < C8) {                            // threads t, t + C8, ... hold the same eight channels (256 % C8 == 0)
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = threadIdx.x; t < 256; t += C8)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += sh[t][k];
    float* o = dbias + (size_t)blockIdx.x * C + threadIdx.x * 8;
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]); *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}
void unpool_relu_backward(hipStream_t s, const float* dpooled, const float* pooled, const uint8_t* idx, float* dy, int B, int Ho,
                          int Wo, int C, int pool, bf16_t* dyb, float* dbias, float* partial, const bf16_t* pooledb, ColsumJobs* defer, const bf16_t* dpooled16) {
  int Hp = Ho / 2, Wp = pool == 1 ? Wo / 2 : Wo;
  if ((Ho & 1) || (pool == 1 && (Wo & 1))) {                                      // floor-mode leftovers
    if (dy) hipMemsetAsync(dy, 0, (size_t)B * Ho * Wo * C * sizeof(float), s);
    if (dyb) hipMemsetAsync(dyb, 0, (size_t)B * Ho * Wo * C * sizeof(bf16_t), s);
  }
  int64_t total = (int64_t)B * Hp * Wp * (C / 4);
  const int C4 = C / 4;
  if (dbias && partial && dyb && (256 % C4 == 0)) {     // fused bias gradient: per-workgroup partial rows, then one small column sum
    int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);             // grid stride 2048*256 is a multiple of every C4 | 256
    if (!dy && pooledb && C % 8 == 0 && 256 % (C / 8) == 0 && !getenv("AOCR_UNPOOL4")) {
      const int64_t total8 = total / 2; const int blocks8 = (int)std::min<int64_t>((total8 + 255) / 256, 2048);
      hipLaunchKernelGGL(unpool8_kernel, dim3(blocks8), dim3(256), 0, s, dpooled, idx, dyb, partial, B, Ho, Wo, C, pool, Hp, Wp, pooledb, dpooled16);
      if (defer) colsum_defer(*defer, partial, C, blocks8, C, dbias); else colsum_accum(s, partial, C, blocks8, C, dbias);
      return;
    }
    if (dy) hipLaunchKernelGGL((unpool_kernel<true, true>), dim3(blocks), dim3(256), 0, s, dpooled, pooled, idx, dy, dyb, partial, B, Ho, Wo, C, pool, Hp, Wp, pooledb);
    else    hipLaunchKernelGGL((unpool_kernel<false, true>), dim3(blocks), dim3(256), 0, s, dpooled, pooled, idx, dy, dyb, partial, B, Ho, Wo, C, pool, Hp, Wp, pooledb);
    if (defer) colsum_defer(*defer, partial, C, blocks, C, dbias); else colsum_accum(s, partial, C, blocks, C, dbias);
    return;
  }
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL((unpool_kernel<true, false>), dim3(blocks), dim3(256), 0, s, dpooled, pooled, idx, dy, dyb, nullptr, B, Ho, Wo, C, pool, Hp, Wp, pooledb);
}

// =============================================================================================
// BatchNorm (+ReLU).  Statistics are accumulated in fp64 (Torch7 accumulates in accreal=double [upstream]).
// scratch layout: double part[BN_CHUNKS][C][2], then double fin[C][2].
// =============================================================================================
static const int BN_CHUNKS = 512;
size_t bn_scratch_bytes(int C) { return (size_t)(BN_CHUNKS + 1) * C * 2 * sizeof(double) + 64; }   // + the row count that travels with the synchronised sums

// mode 0: (sum x, sum x^2);  mode 1: (sum dy, sum dy*xhat) with dy = dA*(y>0), xhat = (x-mean)*invstd
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ dA, const float* __restrict__ save,
                                                         double* __restrict__ part, int64_t rows, int C, int mode,
                                                         int tb_rows, int T, const bf16_t* __restrict__ yb) {
  __shared__ double sh[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t per = (rows + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    float mean = 0.f, inv = 0.f;
    if (mode == 1) { mean = save[c]; inv = save[C + c]; }
    int64_t r = r0 + rl;
    if (mode == 0) {
      double t0 = 0.0, t1 = 0.0;
      for (; r + 4 < r1; r += 8) {                              // two independent rows per iteration
        float xa = x[r * C + c], xb2 = x[(r + 4) * C + c];
        s0 += (double)xa; s1 += (double)xa * (double)xa; t0 += (double)xb2; t1 += (double)xb2 * (double)xb2;
      }
      s0 += t0; s1 += t1;
    }
    for (; r < r1; r += 4) {
      float xv = x[r * C + c];
      if (mode == 0) { s0 += (double)xv; s1 += (double)xv * (double)xv; }
      else {
        int64_t ro = r;
        if (tb_rows > 0) { int64_t b = r / T, t = r - b * T; ro = t * tb_rows + b; }
        float yy = yb ? (float)yb[ro * C + c] : y[ro * C + c];          // only the sign matters (ReLU mask): the bf16 shadow has it
        float d = yy > 0.f ? dA[ro * C + c] : 0.f;
        s0 += (double)d; s1 += (double)d * (double)((xv - mean) * inv);
      }
    }
  }
  sh[0][rl][cl] = s0; sh[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < C) {
    double a = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
    double b = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
    part[((int64_t)blockIdx.y * C + c) * 2] = a; part[((int64_t)blockIdx.y * C + c) * 2 + 1] = b;
  }
}

// The same partial sums with 16-byte accesses and four rows in flight per thread (C / 4 a power of two <= 256): one thread =
// one channel quad, 1024 threads = 1024 / (C/4) rows per pass; the row groups are combined through LDS one value at a time.
// (bn_partial_kernel: 4-byte accesses, one or two loads in flight per thread -- 2.9 TB/s on the 268 MB backward pass of conv3.)
// x of a BatchNorm either as fp32 or (xh != nullptr: the conv epilogue wrote the pre-BatchNorm map as bf16, EpConv::y16) as bf16
__device__ __forceinline__ float4 bn_ldx4(const float* __restrict__ x, const bf16_t* __restrict__ xh, int64_t off) {
  if (xh) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(xh + off); return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]); }
  return *reinterpret_cast<const float4*>(x + off);
}
template <int MODE>
__global__ __launch_bounds__(1024) void bn_partial4_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dA, const float* __restrict__ save,
                                                           double* __restrict__ part, int64_t rows, int C, int tb_rows, int T,
                                                           const bf16_t* __restrict__ yb, const bf16_t* __restrict__ xh = nullptr,
                                                           const bf16_t* __restrict__ dAh = nullptr /* d A as bf16 (conv_backward_data wrote it so: round 4) */) {
  __shared__ double sh[1024];
  const int C4 = C >> 2, q = threadIdx.x & (C4 - 1), rl = threadIdx.x / C4, RP = 1024 / C4, c = q * 4;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
  double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
  float mean[4] = {0.f, 0.f, 0.f, 0.f}, inv[4] = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) {
    const float4 m4 = *reinterpret_cast<const float4*>(save + c), i4 = *reinterpret_cast<const float4*>(save + C + c);
    mean[0] = m4.x; mean[1] = m4.y; mean[2] = m4.z; mean[3] = m4.w; inv[0] = i4.x; inv[1] = i4.y; inv[2] = i4.z; inv[3] = i4.w;
  }
  auto orow = [&](int64_t r) -> int64_t {                     // row of y / dA: (T, B) order for the last layer
    if (tb_rows > 0) { const int64_t b = r / T, t = r - b * T; return t * tb_rows + b; }
    return r;
  };
  auto accum = [&](const float4& xv, const float4& dv, const float (&yy)[4]) {
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ds[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (MODE == 0) { s0[i] += (double)xs[i]; s1[i] += (double)xs[i] * (double)xs[i]; }
      else { const float d = yy[i] > 0.f ? ds[i] : 0.f; s0[i] += (double)d; s1[i] += (double)d * (double)((xs[i] - mean[i]) * inv[i]); }
    }
  };
  auto loady = [&](int64_t ro, float (&yy)[4]) {              // only the sign matters (ReLU mask): the bf16 shadow has it
    if (yb) { const bf16x4 h = *reinterpret_cast<const bf16x4*>(yb + ro * C + c); yy[0] = (float)h[0]; yy[1] = (float)h[1]; yy[2] = (float)h[2]; yy[3] = (float)h[3]; }
    else { const float4 v = *reinterpret_cast<const float4*>(y + ro * C + c); yy[0] = v.x; yy[1] = v.y; yy[2] = v.z; yy[3] = v.w; }
  };
  int64_t r = r0 + rl;
  for (; r + 3 * RP < r1; r += 4 * RP) {
    float4 xv[4], dv[4]; float yy[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xv[u] = bn_ldx4(x, xh, (r + u * RP) * C + c);
      if (MODE == 1) { const int64_t ro = orow(r + u * RP); dv[u] = bn_ldx4(dA, dAh, ro * C + c); loady(ro, yy[u]); }
      else dv[u] = xv[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) accum(xv[u], dv[u], yy[u]);
  }
  for (; r < r1; r += RP) {
    const float4 xv = bn_ldx4(x, xh, r * C + c); float4 dv = xv; float yy[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) { const int64_t ro = orow(r); dv = bn_ldx4(dA, dAh, ro * C + c); loady(ro, yy); }
    accum(xv, dv, yy);
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {                               // combine the RP row groups, one of the 8 values per pass
    sh[threadIdx.x] = v < 4 ? s0[v] : s1[v - 4];
    __syncthreads();
    if (rl == 0) {
      double a = 0.0;
      for (int j = 0; j < RP; ++j) a += sh[j * C4 + q];
      part[((int64_t)blockIdx.x * C + c + (v & 3)) * 2 + (v >> 2)] = a;
    }
    __syncthreads();
  }
}
static bool bn_partial4_ok(int C) { const int C4 = C >> 2; return C % 4 == 0 && C4 >= 1 && C4 <= 256 && (C4 & (C4 - 1)) == 0 && !getenv("AOCR_BN_PARTIAL_OLD"); }

// sums the per-chunk partials of 16 channels with 256 threads (16 k-slices per channel, LDS tree): returns the totals to the
// 16 threads with kslice == 0
__device__ __forceinline__ bool bn_reduce_partials(const double* __restrict__ part, int nchunk, int C, int& c, double& s, double& ss) {
  __shared__ double sh[2][16][17];
  const int cl = threadIdx.x & 15, ks = threadIdx.x >> 4;
  c = blockIdx.x * 16 + cl;
  double a = 0, b = 0;
  if (c < C) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    int k = ks;
    for (; k + 7 * 16 < nchunk; k += 8 * 16) {                    // eight 16-byte loads in flight (a serial chain of 32 round trips cost ~10 us per launch)
      d2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const d2*>(part + ((int64_t)(k + 16 * u) * C + c) * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) { a += v[u][0]; b += v[u][1]; }
    }
    for (; k < nchunk; k += 16) { a += part[((int64_t)k * C + c) * 2]; b += part[((int64_t)k * C + c) * 2 + 1]; }
  }
  sh[0][ks][cl] = a; sh[1][ks][cl] = b;
  __syncthreads();
  if (ks != 0 || c >= C) return false;
  s = 0; ss = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { s += sh[0][k][cl]; ss += sh[1][k][cl]; }
  return true;
}
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const double* __restrict__ part, int nchunk, int64_t rows, int C, float* save,
                                                              float* rm, float* rv, int update_running) {
  int c; double s, ss;
  if (!bn_reduce_partials(part, nchunk, C, c, s, ss)) return;
  double n = (double)rows, mean = s / n, var = ss / n - mean * mean;
  if (var < 0) var = 0;
  save[c] = (float)mean; save[C + c] = (float)(1.0 / sqrt(var + 1e-5));
  if (update_running) {                                         // momentum 0.1, unbiased variance [upstream THNN BatchNormalization]
    double unb = rows > 1 ? var * n / (n - 1.0) : var;
    rm[c] = (float)(0.1 * mean + 0.9 * (double)rm[c]);
    rv[c] = (float)(0.1 * unb + 0.9 * (double)rv[c]);
  }
}
// ---- synchronised BatchNorm (data parallelism): the per-channel sums leave the reduction un-normalised, travel through ONE
// all-reduce together with the row count (fin[2C]), and are normalised by the global count afterwards.
__global__ __launch_bounds__(256) void bn_sums_kernel(const double* __restrict__ part, int nchunk, int64_t rows, int C, double* fin, float* dw,
                                                      float* db) {
  int c; double s, ss;
  if (blockIdx.x == 0 && threadIdx.x == 0) fin[2 * C] = (double)rows;
  if (!bn_reduce_partials(part, nchunk, C, c, s, ss)) return;
  fin[c * 2] = s; fin[c * 2 + 1] = ss;
  if (dw) { dw[c] += (float)ss; db[c] += (float)s; }                       // backward: the LOCAL sums (the gradient exchange adds the ranks up)
}
__global__ void bn_fwd_stats_kernel(const double* __restrict__ fin, int C, float* save, float* rm, float* rv, int update_running) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double n = fin[2 * C], mean = fin[c * 2] / n; double var = fin[c * 2 + 1] / n - mean * mean;
  if (var < 0) var = 0;
  save[c] = (float)mean; save[C + c] = (float)(1.0 / sqrt(var + 1e-5));
  if (update_running) {
    const double unb = n > 1 ? var * n / (n - 1.0) : var;
    rm[c] = (float)(0.1 * mean + 0.9 * (double)rm[c]);
    rv[c] = (float)(0.1 * unb + 0.9 * (double)rv[c]);
  }
}
__global__ void bn_bwd_scale_kernel(double* fin, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * C) fin[i] = fin[i] / fin[2 * C];
}
__global__ void bn_eval_prepare_kernel(const float* rm, const float* rv, float* save, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { save[c] = rm[c]; save[C + c] = (float)(1.0 / sqrt((double)rv[c] + 1e-5)); }
}
void bn_eval_prepare(hipStream_t s, const float* rm, const float* rv, float* save, int C) {
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, rm, rv, save, C);
}
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ save, int64_t rows, int C, int tb_rows,
                                                            int T, bf16_t* __restrict__ yb, const bf16_t* __restrict__ xh = nullptr) {
  const int C4 = C >> 2;
  const int64_t total = rows * C4;
  for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
    int c = (int)(id % C4) * 4; int64_t r = id / C4;
    float4 xv = bn_ldx4(x, xh, r * C + c);
    float4 m = *reinterpret_cast<const float4*>(save + c), iv = *reinterpret_cast<const float4*>(save + C + c);
    float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(b + c);
    float4 o;
    o.x = fmaxf((xv.x - m.x) * iv.x * ww.x + bb.x, 0.f); o.y = fmaxf((xv.y - m.y) * iv.y * ww.y + bb.y, 0.f);
    o.z = fmaxf((xv.z - m.z) * iv.z * ww.z + bb.z, 0.f); o.w = fmaxf((xv.w - m.w) * iv.w * ww.w + bb.w, 0.f);
    int64_t ro = r;
    if (tb_rows > 0) { int64_t bi = r / T, t = r - bi * T; ro = t * tb_rows + bi; }
    if (y) *reinterpret_cast<float4*>(y + ro * C + c) = o;        // y == nullptr: only the bf16 shadow is kept (bf16 mode, inner layers)
    if (yb) { bf16x4 hb; hb[0] = (bf16_t)o.x; hb[1] = (bf16_t)o.y; hb[2] = (bf16_t)o.z; hb[3] = (bf16_t)o.w; *reinterpret_cast<bf16x4*>(yb + ro * C + c) = hb; }
  }
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ part, int nchunk, int64_t rows, int C,
                                                              const float* save, double* fin, float* dw, float* db) {
  int c; double s, ss;
  if (!bn_reduce_partials(part, nchunk, C, c, s, ss)) return;
  fin[c * 2] = s / (double)rows; fin[c * 2 + 1] = ss / (double)rows;
  dw[c] += (float)ss; db[c] += (float)s;                        // gradWeight = sum dy*xhat, gradBias = sum dy
}
// F32OUT = false / BIAS = true (bf16 mode): only the bf16 shadow of dx is written (every consumer reads the shadow) and the
// gradient of the preceding conv's bias, sum_rows dx[., c], is accumulated per thread (the grid stride is a multiple of C, so a
// thread keeps its channel) into a flat partial slab that a small column sum finishes.
template <bool F32OUT, bool BIAS>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ dA, const float* __restrict__ w,
                                                           const float* __restrict__ save, const double* __restrict__ fin,
                                                           float* __restrict__ dx, int64_t rows, int C, int tb_rows, int T,
                                                           bf16_t* __restrict__ dxb, const bf16_t* __restrict__ yb,
                                                           float* __restrict__ partial, const bf16_t* __restrict__ xh = nullptr, const bf16_t* __restrict__ dAh = nullptr) {
  const int C4 = C >> 2;                                // one channel quad per thread and iteration (16-byte accesses)
  const int64_t total = rows * C4;
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  // rows are walked from the END: the partial-sum pass that ran just before read x / dA / y front to back (301 MB at conv3: more than the
  // 256 MB Infinity Cache), so its tail is what is still cached
  const int64_t stride = (int64_t)gridDim.x * 256, first = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t niter = first < total ? (total - 1 - first) / stride + 1 : 0;
  for (int64_t it = niter - 1; it >= 0; --it) {
    const int64_t id = first + it * stride;
    const int c = (int)(id % C4) * 4; const int64_t r = id / C4;
    int64_t ro = r;
    if (tb_rows > 0) { int64_t bi = r / T, t = r - bi * T; ro = t * tb_rows + bi; }
    float yy[4];
    if (yb) { bf16x4 t4 = *reinterpret_cast<const bf16x4*>(yb + ro * C + c); yy[0] = (float)t4[0]; yy[1] = (float)t4[1]; yy[2] = (float)t4[2]; yy[3] = (float)t4[3]; }
    else { float4 t4 = *reinterpret_cast<const float4*>(y + ro * C + c); yy[0] = t4.x; yy[1] = t4.y; yy[2] = t4.z; yy[3] = t4.w; }
    const float4 da4 = bn_ldx4(dA, dAh, ro * C + c);
    const float4 x4 = bn_ldx4(x, xh, r * C + c);
    const float da[4] = {da4.x, da4.y, da4.z, da4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
    float gx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = yy[k] > 0.f ? da[k] : 0.f;
      const float inv = save[C + c + k];
      const float xh = (xv[k] - save[c + k]) * inv;
      gx[k] = (d - (float)fin[(c + k) * 2] - xh * (float)fin[(c + k) * 2 + 1]) * inv * w[c + k];
      if (BIAS) bsum[k] += gx[k];
    }
    if (F32OUT) *reinterpret_cast<float4*>(dx + r * C + c) = make_float4(gx[0], gx[1], gx[2], gx[3]);
    if (dxb) { bf16x4 hb; hb[0] = (bf16_t)gx[0]; hb[1] = (bf16_t)gx[1]; hb[2] = (bf16_t)gx[2]; hb[3] = (bf16_t)gx[3]; *reinterpret_cast<bf16x4*>(dxb + r * C + c) = hb; }
  }
  if (BIAS) *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
}

void bn_relu_forward(hipStream_t s, const float* x, float* y, const float* w, const float* b, float* rm, float* rv,
                     float* save, void* scratch, int64_t rows, int C, int training, int update_running, int tb_rows, bf16_t* yb,
                     const BnSync* sync, int stats_chunks, const bf16_t* xh) {
  int T = tb_rows > 0 ? (int)(rows / tb_rows) : 0;
  if (training) {
    double* part = (double*)scratch;
    int nchunk = (int)std::min<int64_t>(BN_CHUNKS, (rows + 63) / 64);
    if (stats_chunks > 0 && stats_chunks <= BN_CHUNKS) nchunk = stats_chunks;      // the producing conv's epilogue wrote the partial sums (EpConv::bn_part)
    else
    if (bn_partial4_ok(C)) hipLaunchKernelGGL(bn_partial4_kernel<0>, dim3(nchunk), dim3(1024), 0, s, x, nullptr, nullptr, nullptr, part, rows, C, 0, 0, nullptr);
    else hipLaunchKernelGGL(bn_partial_kernel, dim3(cdiv(C, 64), nchunk), dim3(256), 0, s, x, nullptr, nullptr, nullptr, part, rows,
                            C, 0, 0, 0, nullptr);
    if (sync) {                                           // statistics of the GLOBAL batch: (sum x, sum x^2, rows) summed over the ranks
      double* fin = part + (size_t)BN_CHUNKS * C * 2;
      hipLaunchKernelGGL(bn_sums_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, part, nchunk, rows, C, fin, nullptr, nullptr);
      sync->allreduce(sync->ctx, fin, 2 * C + 1, 1, s);
      hipLaunchKernelGGL(bn_fwd_stats_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, fin, C, save, rm, rv, update_running);
    } else
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, part, nchunk, rows, C, save, rm, rv,
                       update_running);
  } else {
    hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, rm, rv, save, C);
  }
  int64_t total = rows * (C / 4);
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(blocks), dim3(256), 0, s, x, y, w, b, save, rows, C, tb_rows, T, yb, xh);
}
void bn_relu_backward(hipStream_t s, const float* x, const float* y, const float* dA, const float* w, const float* save,
                      float* dx, float* dw, float* db, void* scratch, int64_t rows, int C, int tb_rows, bf16_t* dxb,
                      const bf16_t* yb, float* conv_dbias, float* partial, const BnSync* sync, ColsumJobs* defer, const bf16_t* xh, const bf16_t* dAh, int sums_chunks) {
  int T = tb_rows > 0 ? (int)(rows / tb_rows) : 0;
  double* part = (double*)scratch;
  double* fin = part + (size_t)BN_CHUNKS * C * 2;
  int nchunk = (int)std::min<int64_t>(BN_CHUNKS, (rows + 63) / 64);
  if (sums_chunks > 0 && sums_chunks <= BN_CHUNKS) nchunk = sums_chunks;       // the producing data gradient's epilogue wrote the partial sums (EpStore::bnb_part)
  else
  if (bn_partial4_ok(C)) hipLaunchKernelGGL(bn_partial4_kernel<1>, dim3(nchunk), dim3(1024), 0, s, x, y, dA, save, part, rows, C, tb_rows, T, yb, xh, dAh);
  else hipLaunchKernelGGL(bn_partial_kernel, dim3(cdiv(C, 64), nchunk), dim3(256), 0, s, x, y, dA, save, part, rows, C, 1, tb_rows, T, yb);
  if (sync) {                                             // mean(dy), mean(dy * xhat) over the GLOBAL batch
    hipLaunchKernelGGL(bn_sums_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, part, nchunk, rows, C, fin, dw, db);
    sync->allreduce(sync->ctx, fin, 2 * C + 1, 1, s);
    hipLaunchKernelGGL(bn_bwd_scale_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, s, fin, C);
  } else
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, part, nchunk, rows, C, save, fin, dw, db);
  const int C4 = C / 4;
  int64_t total = rows * C4;
  int blocks = (int)std::min<int64_t>((total + 255) / 256, 16384);
  if (conv_dbias && partial && dxb && 256 % C4 == 0) {  // fused bias gradient of the preceding conv (+ optional fp32 output)
    // the grid stride (blocks * 256 quads) is a multiple of C4, so a thread keeps its channel quad: the flat partial slab
    // [blocks * 256][4] is a [blocks * 256 / C4][C] matrix whose column sums are the bias gradient
    int fb = (int)std::min<int64_t>((total + 255) / 256, 2048);
    if (dx) hipLaunchKernelGGL((bn_bwd_apply_kernel<true, true>), dim3(fb), dim3(256), 0, s, x, y, dA, w, save, fin, dx, rows, C, tb_rows, T, dxb, yb, partial, xh, dAh);
    else    hipLaunchKernelGGL((bn_bwd_apply_kernel<false, true>), dim3(fb), dim3(256), 0, s, x, y, dA, w, save, fin, dx, rows, C, tb_rows, T, dxb, yb, partial, xh, dAh);
    if (defer) colsum_defer(*defer, partial, C, (int64_t)fb * 256 / C4, C, conv_dbias); else colsum_accum(s, partial, C, (int64_t)fb * 256 / C4, C, conv_dbias);
    return;
  }
  hipLaunchKernelGGL((bn_bwd_apply_kernel<true, false>), dim3(blocks), dim3(256), 0, s, x, y, dA, w, save, fin, dx, rows, C, tb_rows, T, dxb, yb, nullptr, xh, dAh);
}

// =============================================================================================
// attention core, LSTM.lua:133-150: one workgroup per batch row.
//   phase 1: r[t] = <ctx[b,t,:], u[b,:]>     (wave per t, lanes over Hd, wave reduction)
//   phase 2: softmax (forward) or softmax-backward (backward) over T in LDS
//   phase 3: o[:] = sum_t p[t] * ctx[b,t,:]  (threads over Hd)
// forward:  u=q, p=a=softmax(r), o=c.   backward: u=dc, r=da, p=ds=a*(da-sum a*da), o=dq.
// =============================================================================================
template <bool BWD>
__global__ __launch_bounds__(256) void attn_core_kernel(const float* __restrict__ ctx, const float* __restrict__ u, int64_t ldu,
                                                        const float* __restrict__ a_in, float* __restrict__ p_out,
                                                        float* __restrict__ o, int64_t ldo, int T, int Hd, int ctx_div,
                                                        bf16_t* __restrict__ ob, int64_t ldob) {
  extern __shared__ float sm[];                // [T] scores + [8] scratch
  float* sc = sm; float* red = sm + T;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cb = ctx + (int64_t)(b / ctx_div) * T * Hd;
  const float* ub = u + (int64_t)b * ldu;
  for (int t = wave; t < T; t += 4) {
    float s = 0.f;
    for (int j = lane * 4; j < Hd; j += 256) {
      float4 cv = *reinterpret_cast<const float4*>(cb + (int64_t)t * Hd + j);
      float4 uv = *reinterpret_cast<const float4*>(ub + j);
      s = fmaf(cv.x, uv.x, s); s = fmaf(cv.y, uv.y, s); s = fmaf(cv.z, uv.z, s); s = fmaf(cv.w, uv.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) sc[t] = s;
  }
  __syncthreads();
  if (!BWD) {
    float m = -INFINITY;
    for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, sc[t]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) { float e = expf(sc[t] - m); sc[t] = e; sum += e; }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    float inv = 1.f / sum;
    for (int t = threadIdx.x; t < T; t += 256) { float a = sc[t] * inv; sc[t] = a; p_out[(int64_t)b * T + t] = a; }
  } else {
    float dot = 0.f;
    for (int t = threadIdx.x; t < T; t += 256) dot += a_in[(int64_t)b * T + t] * sc[t];
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    dot = red[0] + red[1] + red[2] + red[3];
    for (int t = threadIdx.x; t < T; t += 256) {
      float ds = a_in[(int64_t)b * T + t] * (sc[t] - dot); sc[t] = ds; p_out[(int64_t)b * T + t] = ds;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x * 4; j < Hd; j += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      float p = sc[t];
      float4 cv = *reinterpret_cast<const float4*>(cb + (int64_t)t * Hd + j);
      acc.x = fmaf(p, cv.x, acc.x); acc.y = fmaf(p, cv.y, acc.y); acc.z = fmaf(p, cv.z, acc.z); acc.w = fmaf(p, cv.w, acc.w);
    }
    *reinterpret_cast<float4*>(o + (int64_t)b * ldo + j) = acc;
    if (ob) { bf16x4 hb; hb[0] = (bf16_t)acc.x; hb[1] = (bf16_t)acc.y; hb[2] = (bf16_t)acc.z; hb[3] = (bf16_t)acc.w; *reinterpret_cast<bf16x4*>(ob + (int64_t)b * ldob + j) = hb; }
  }
}
// Register-resident variant for T <= 64 and Hd = 256*NC (NC = 1, 2): the whole (T, Hd) context slice of one batch row
// is loaded ONCE into registers (wave w holds rows t = w, w+4, ...; all 16*NC dwordx4 loads of a lane are in flight
// together) and serves both the score pass and the weighted-sum pass.  Same phases and outputs as attn_core_kernel.
__device__ __forceinline__ float4 attn_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 attn_load4(const bf16_t* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
// CT = bf16_t (bf16 compute mode): the context is read from its bf16 shadow -- 16.5 MB instead of 33 MB per launch at C3, which is
// what this kernel's duration is made of (one workgroup per batch row streams its whole (T, Hd) slice), and small enough to stay
// in the XCD's L2 between the 2 x 24 launches of a step (batch row b always lands on XCD b % 8).
template <int NC, bool BWD, class CT>
__global__ __launch_bounds__(256) void attn_reg_kernel(const CT* __restrict__ ctx, const float* __restrict__ u, int64_t ldu,
                                                       const float* __restrict__ a_in, float* __restrict__ p_out,
                                                       float* __restrict__ o, int64_t ldo, int T, int ctx_div, bf16_t* __restrict__ ob,
                                                       int64_t ldob) {
  constexpr int Hd = 256 * NC;
  __shared__ float sc[64];
  __shared__ __attribute__((aligned(16))) float red[4][Hd];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const CT* cb = ctx + (int64_t)(b / ctx_div) * T * Hd;
  const float* ub = u + (int64_t)b * ldu;
  float4 c[16][NC];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int t = wave + 4 * i;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
      c[i][cc] = t < T ? attn_load4(cb + (int64_t)t * Hd + cc * 256 + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 uu[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) uu[cc] = *reinterpret_cast<const float4*>(ub + cc * 256 + lane * 4);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float s = 0.f;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      s = fmaf(c[i][cc].x, uu[cc].x, s); s = fmaf(c[i][cc].y, uu[cc].y, s); s = fmaf(c[i][cc].z, uu[cc].z, s); s = fmaf(c[i][cc].w, uu[cc].w, s);
    }
    s = wave_sum(s);
    if (lane == 0) sc[wave + 4 * i] = s;
  }
  __syncthreads();
  if (wave == 0) {                                             // T <= 64: one wave does the softmax / its backward
    const bool ok = lane < T;
    float v = ok ? sc[lane] : -INFINITY;
    float p;
    if (!BWD) {
      float m = wave_max(v);
      float e = ok ? expf(v - m) : 0.f;
      float sum = wave_sum(e);
      p = e * (1.f / sum);
    } else {
      float a = ok ? a_in[(int64_t)b * T + lane] : 0.f;
      float dot = wave_sum(ok ? a * v : 0.f);
      p = ok ? a * (v - dot) : 0.f;
    }
    sc[lane] = p;
    if (ok) p_out[(int64_t)b * T + lane] = p;
  }
  __syncthreads();
  float4 acc[NC];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) acc[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float p = sc[wave + 4 * i];                          // rows >= T carry p = 0 and c = 0
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      acc[cc].x = fmaf(p, c[i][cc].x, acc[cc].x); acc[cc].y = fmaf(p, c[i][cc].y, acc[cc].y);
      acc[cc].z = fmaf(p, c[i][cc].z, acc[cc].z); acc[cc].w = fmaf(p, c[i][cc].w, acc[cc].w);
    }
  }
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) *reinterpret_cast<float4*>(&red[wave][cc * 256 + lane * 4]) = acc[cc];
  __syncthreads();
  for (int j = threadIdx.x; j < Hd; j += 256) {
    const float v = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    o[(int64_t)b * ldo + j] = v;
    if (ob) ob[(int64_t)b * ldob + j] = (bf16_t)v;
  }
}

// bf16-context form of attn_reg_kernel for Hd = 512: one 16-byte load per lane and context row (8 bf16), so the (T, 512) slice
// of a batch row is 16 load instructions per lane instead of 32 -- the kernel is bound by load issue/latency, not by bytes
// (reading the bf16 shadow with 8-byte loads was slower than fp32: 15.5 vs 12.2 us).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <bool BWD, int NW>
__global__ __launch_bounds__(64 * NW) void attn_reg_h512_kernel(const bf16_t* __restrict__ ctx, const float* __restrict__ u, int64_t ldu,
                                                            const float* __restrict__ a_in, float* __restrict__ p_out,
                                                            float* __restrict__ o, int64_t ldo, int T, int ctx_div,
                                                            bf16_t* __restrict__ ob, int64_t ldob) {
  constexpr int Hd = 512, RW = 64 / NW;                        // RW context rows per wave
  __shared__ float sc[64];
  __shared__ __attribute__((aligned(16))) float red[NW][Hd];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* cb = ctx + (int64_t)(b / ctx_div) * T * Hd + lane * 8;
  const float* ub = u + (int64_t)b * ldu + lane * 8;
  bf16x8 c[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int t = wave + NW * i;
    if (t < T) c[i] = *reinterpret_cast<const bf16x8*>(cb + (int64_t)t * Hd);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) c[i][e] = (bf16_t)0.f;
    }
  }
  const float4 u0 = *reinterpret_cast<const float4*>(ub), u1 = *reinterpret_cast<const float4*>(ub + 4);
  const float uu[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf((float)c[i][e], uu[e], s);
    s = wave_sum(s);
    if (lane == 0) sc[wave + NW * i] = s;
  }
  __syncthreads();
  if (wave == 0) {                                             // T <= 64: one wave does the softmax / its backward
    const bool ok = lane < T;
    float v = ok ? sc[lane] : -INFINITY;
    float p;
    if (!BWD) {
      float m = wave_max(v);
      float e = ok ? expf(v - m) : 0.f;
      float sum = wave_sum(e);
      p = e * (1.f / sum);
    } else {
      float a = ok ? a_in[(int64_t)b * T + lane] : 0.f;
      float dot = wave_sum(ok ? a * v : 0.f);
      p = ok ? a * (v - dot) : 0.f;
    }
    sc[lane] = p;
    if (ok) p_out[(int64_t)b * T + lane] = p;
  }
  __syncthreads();
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const float p = sc[wave + NW * i];                          // rows >= T carry p = 0 and c = 0
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, (float)c[i][e], acc[e]);
  }
  *reinterpret_cast<float4*>(&red[wave][lane * 8]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *reinterpret_cast<float4*>(&red[wave][lane * 8 + 4]) = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  for (int j = threadIdx.x; j < Hd; j += 64 * NW) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < NW; q += 4) v += (red[q][j] + red[q + 1][j]) + (red[q + 2][j] + red[q + 3][j]);
    o[(int64_t)b * ldo + j] = v;
    if (ob) ob[(int64_t)b * ldob + j] = (bf16_t)v;
  }
}

// General-T form of the bf16-context kernel, Hd = 512 * NC (He = 256: BASELINE C3/C4; He = 512: the reference default and C5).
// 16 waves per batch row, wave w owns context rows t = w, w+16, ...  STREAM = false: the row's whole (T <= 16 RW, Hd) slice sits in
// registers (RW x NC 16-byte fragments per lane) and serves the score pass and the weighted-sum pass -- one read of the context, as in
// attn_reg_h512_kernel, for T up to 256 (C4's widest bucket is T = 199).  STREAM = true: any T (C5: T = 1785), two passes over the
// bf16 shadow in chunks of 16 RW rows, scores parked in LDS in between.  Softmax / its backward over T by wave 0.
// NW (round 6): 8 waves for T <= 32 at Hd = 1024 (the reference-default decoder: T = 24) -- half the barrier population and half the 64 KB reduction image of the
// 16-wave form, twice the workgroups per CU: reference-default step 8.42 -> 8.26 ms (48 calls per step; four waves of 8 rows measured slower: 8.32).  AOCR_ATTN_NW16=1: the 16-wave form.
// DUAL (round 6, the Hd = 1024 launch chain at T <= 64): scores against the PRE-MULTIPLIED context, as the whole-sequence kernels do -- ctx[t] . (W_a h) =
// (ctx W_a)[t] . h (LSTM.lua:131-137) -- so the chain has no q = W_a h launch per step and the second cell launch of the backward step no K = Hd product:
// forward: scores from ctx2 = bf16(ctx W_a) against u = h_top, weighted sum from ctx; backward: scores (d a) and d q from ctx as before, and a SECOND weighted
// sum of the same d s over ctx2 = d h_top's attention part, written to o2.  The context is read twice per call (19.7 MB each at the reference-default shape).
template <bool BWD, int NC, int RW, bool STREAM, int NW = 16, bool DUAL = false>
__global__ __launch_bounds__(64 * NW) void attn_bf16_kernel(const bf16_t* __restrict__ ctx, const float* __restrict__ u, int64_t ldu,
                                                         const float* __restrict__ a_in, float* __restrict__ p_out,
                                                         float* __restrict__ o, int64_t ldo, int T, int ctx_div,
                                                         bf16_t* __restrict__ ob, int64_t ldob,
                                                         const bf16_t* __restrict__ ctx2 = nullptr, float* __restrict__ o2 = nullptr, int64_t ldo2 = 0,
                                                         const float* __restrict__ cfwd = nullptr, int64_t ldcf = 0) {
  static_assert(!(DUAL && STREAM), "the two-context form keeps a row's slice in registers");
  constexpr int Hd = 512 * NC;
  extern __shared__ float sc[];                                 // T scores / probabilities
  __shared__ __attribute__((aligned(16))) float red[NW][Hd];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* cb = ctx + (int64_t)(b / ctx_div) * T * Hd + lane * 8;
  const bf16_t* cb2 = DUAL ? ctx2 + (int64_t)(b / ctx_div) * T * Hd + lane * 8 : cb;
  const float* ub = u + (int64_t)b * ldu + lane * 8;
  float uu[NC][8];
#pragma unroll
  for (int cc = 0; cc < NC; ++cc) {
    const float4 u0 = *reinterpret_cast<const float4*>(ub + cc * 512), u1 = *reinterpret_cast<const float4*>(ub + cc * 512 + 4);
    uu[cc][0] = u0.x; uu[cc][1] = u0.y; uu[cc][2] = u0.z; uu[cc][3] = u0.w; uu[cc][4] = u1.x; uu[cc][5] = u1.y; uu[cc][6] = u1.z; uu[cc][7] = u1.w;
  }
  bf16x8 c[RW][NC];
  auto load_rows = [&](const bf16_t* base, int t0) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + wave + NW * i;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        if (t < T) c[i][cc] = *reinterpret_cast<const bf16x8*>(base + (int64_t)t * Hd + cc * 512);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) c[i][cc][e] = (bf16_t)0.f;
        }
      }
    }
  };
  auto scores = [&](int t0) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      float s = 0.f;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf((float)c[i][cc][e], uu[cc][e], s);
      s = wave_sum(s);
      const int t = t0 + wave + NW * i;
      if (lane == 0 && t < T) sc[t] = s;
    }
  };
  if constexpr (STREAM && BWD) {
    if (cfwd) {
      // Round 6: ONE pass for the streamed BACKWARD form too.  d q = sum_t a_t (d a_t - dot) ctx_t with dot = sum_t a_t d a_t needs dot before the weighted sum -- unless
      // it is split: d q = [sum_t a_t d a_t ctx_t] - dot [sum_t a_t ctx_t], and the second bracket is the FORWARD pass's weighted context (cfwd: the c half of the saved
      // [c ; h_top]).  So one sweep computes d a_t = ctx_t . d c, keeps it in sc[] for d s, and accumulates w_t = a_t d a_t into dot and w_t ctx_t into G.
      // (G and dot cfwd nearly cancel where the attention is peaked: an absolute error of ~1e-7 |d a| |c|, far below the bf16 rounding of the context rows themselves.)
      float dot_run = 0.f, g1[NC][8];
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) g1[cc][e] = 0.f;
      for (int t0 = 0; t0 < T; t0 += NW * RW) {
        load_rows(cb, t0);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
          const int t = t0 + wave + NW * i;
          if (t < T) {                                            // (wave-uniform)
            float da = 0.f;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc)
#pragma unroll
              for (int e = 0; e < 8; ++e) da = fmaf((float)c[i][cc][e], uu[cc][e], da);
            da = wave_sum(da);
            if (lane == 0) sc[t] = da;
            const float w = a_in[(int64_t)b * T + t] * da;
            dot_run += w;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc)
#pragma unroll
              for (int e = 0; e < 8; ++e) g1[cc][e] = fmaf(w, (float)c[i][cc][e], g1[cc][e]);
          }
        }
      }
      float* const dl = &red[0][0];                             // [NW]: the waves' shares of dot
      if (lane == 0) dl[wave] = dot_run;
      __syncthreads();
      float dot = 0.f;
#pragma unroll
      for (int q = 0; q < NW; ++q) dot += dl[q];
      for (int t = threadIdx.x; t < T; t += 64 * NW) p_out[(int64_t)b * T + t] = a_in[(int64_t)b * T + t] * (sc[t] - dot);
      __syncthreads();                                          // every thread has read the shares before the image is overwritten
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8]) = make_float4(g1[cc][0], g1[cc][1], g1[cc][2], g1[cc][3]);
        *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8 + 4]) = make_float4(g1[cc][4], g1[cc][5], g1[cc][6], g1[cc][7]);
      }
      __syncthreads();
      for (int j = threadIdx.x; j < Hd; j += 64 * NW) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < NW; q += 4) v += (red[q][j] + red[q + 1][j]) + (red[q + 2][j] + red[q + 3][j]);
        v = fmaf(-dot, cfwd[(int64_t)b * ldcf + j], v);
        o[(int64_t)b * ldo + j] = v;
        if (ob) ob[(int64_t)b * ldob + j] = (bf16_t)v;
      }
      return;
    }
  }
  if constexpr (STREAM && !BWD) {
    // Round 6: ONE pass over the context for the streamed forward form (C5: T = 1785, 0.94 GB of context per decoder step at 256 strips -- the two passes were
    // most of that shape's decoder forward and decode time).  Online softmax: every wave keeps a running maximum m, the sum l of e^(s - m) and the weighted sum of
    // its rows, rescaled when the maximum moves; the waves' (m, l) meet in LDS, each wave scales its accumulator by e^(m_w - M) / L and the usual cross-wave sum
    // follows.  The raw scores still go through sc[] so that the probabilities (saved for the backward pass) are e^(s - M) / L of the same M, L.
    float m_run = -INFINITY, l_run = 0.f, acc1[NC][8];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc1[cc][e] = 0.f;
    for (int t0 = 0; t0 < T; t0 += NW * RW) {
      load_rows(cb, t0);
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        const int t = t0 + wave + NW * i;
        if (t < T) {                                              // (wave-uniform)
          float sdot = 0.f;
#pragma unroll
          for (int cc = 0; cc < NC; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf((float)c[i][cc][e], uu[cc][e], sdot);
          sdot = wave_sum(sdot);
          if (lane == 0) sc[t] = sdot;
          const float m_new = fmaxf(m_run, sdot), scale = __expf(m_run - m_new), pe = __expf(sdot - m_new);
          l_run = l_run * scale + pe; m_run = m_new;
#pragma unroll
          for (int cc = 0; cc < NC; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc1[cc][e] = fmaf(pe, (float)c[i][cc][e], acc1[cc][e] * scale);
        }
      }
    }
    float* const ml = &red[0][0];                               // [NW][2] = (m_w, l_w): the reduction image is idle until reduce_store
    if (lane == 0) { ml[2 * wave] = m_run; ml[2 * wave + 1] = l_run; }
    __syncthreads();
    float M = -INFINITY, Lsum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) M = fmaxf(M, ml[2 * q]);
#pragma unroll
    for (int q = 0; q < NW; ++q) Lsum += ml[2 * q + 1] * __expf(ml[2 * q] - M);      // (a wave without rows: l = 0, e^(-inf) = 0)
    const float inv = 1.f / Lsum, wsc = __expf(m_run - M) * inv;
    for (int t = threadIdx.x; t < T; t += 64 * NW) p_out[(int64_t)b * T + t] = __expf(sc[t] - M) * inv;
    __syncthreads();                                            // every thread has read (m, l) before the image is overwritten
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8]) = make_float4(acc1[cc][0] * wsc, acc1[cc][1] * wsc, acc1[cc][2] * wsc, acc1[cc][3] * wsc);
      *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8 + 4]) = make_float4(acc1[cc][4] * wsc, acc1[cc][5] * wsc, acc1[cc][6] * wsc, acc1[cc][7] * wsc);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Hd; j += 64 * NW) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW; q += 4) v += (red[q][j] + red[q + 1][j]) + (red[q + 2][j] + red[q + 3][j]);
      o[(int64_t)b * ldo + j] = v;
      if (ob) ob[(int64_t)b * ldob + j] = (bf16_t)v;
    }
    return;
  }
  if (!STREAM) { load_rows((DUAL && !BWD) ? cb2 : cb, 0); scores(0); }
  else for (int t0 = 0; t0 < T; t0 += NW * RW) { load_rows(cb, t0); scores(t0); }
  __syncthreads();
  if (wave == 0) {                                              // softmax (forward) / softmax backward over T
    if (!BWD) {
      float m = -INFINITY;
      for (int t = lane; t < T; t += 64) m = fmaxf(m, sc[t]);
      m = wave_max(m);
      float sum = 0.f;
      for (int t = lane; t < T; t += 64) { const float e = expf(sc[t] - m); sc[t] = e; sum += e; }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      for (int t = lane; t < T; t += 64) { const float p = sc[t] * inv; sc[t] = p; p_out[(int64_t)b * T + t] = p; }
    } else {
      float dot = 0.f;
      for (int t = lane; t < T; t += 64) dot += a_in[(int64_t)b * T + t] * sc[t];
      dot = wave_sum(dot);
      for (int t = lane; t < T; t += 64) { const float p = a_in[(int64_t)b * T + t] * (sc[t] - dot); sc[t] = p; p_out[(int64_t)b * T + t] = p; }
    }
  }
  if (DUAL && !BWD) load_rows(cb, 0);                           // (issued in front of the barrier: the rows arrive under wave 0's softmax)
  __syncthreads();
  float acc[NC][8];
  auto accumulate = [&](int t0) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + wave + NW * i;
      const float p = t < T ? sc[t] : 0.f;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[cc][e] = fmaf(p, (float)c[i][cc][e], acc[cc][e]);
    }
  };
  auto reduce_store = [&](float* oo, int64_t ld, bf16_t* oob, int64_t ldb) {
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8]) = make_float4(acc[cc][0], acc[cc][1], acc[cc][2], acc[cc][3]);
      *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8 + 4]) = make_float4(acc[cc][4], acc[cc][5], acc[cc][6], acc[cc][7]);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Hd; j += 64 * NW) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NW; q += 4) v += (red[q][j] + red[q + 1][j]) + (red[q + 2][j] + red[q + 3][j]);
      oo[(int64_t)b * ld + j] = v;
      if (oob) oob[(int64_t)b * ldb + j] = (bf16_t)v;
    }
  };
#pragma unroll
  for (int cc = 0; cc < NC; ++cc)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[cc][e] = 0.f;
  if (!STREAM) accumulate(0);
  else for (int t0 = 0; t0 < T; t0 += NW * RW) { load_rows(cb, t0); accumulate(t0); }
  if (DUAL && BWD) load_rows(cb2, 0);                           // the second weighted sum's rows, requested under the first one's reduction
  reduce_store(o, ldo, ob, ldob);
  if (DUAL && BWD) {
    __syncthreads();                                            // every thread is done with the reduction image
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[cc][e] = 0.f;
    accumulate(0);
    reduce_store(o2, ldo2, nullptr, 0);
  }
}

// Beam decode at long T (round 6; BASELINE config 5: beam 5 over T = 1785 positions): the k hypotheses of an image attend to the SAME context rows
// (model.lua:373: the context is not replicated), but attn_bf16_kernel<..., STREAM> runs one workgroup per hypothesis, so every context row is read 2 k times per
// step (9.4 GB per step at 256 strips).  Here ONE workgroup per image: a chunk of rows is loaded once per pass and scored against / accumulated for all k
// hypotheses (k <= KB query vectors and KB accumulators in registers: 8 waves, <= 256 VGPRs); softmax of row j by wave j; the cross-wave sums of the k outputs go
// through the same LDS image one after the other.  Forward only (decode has no backward).  Same arithmetic per hypothesis as the per-row kernel up to the order
// in which the 8 (not 16) waves' partial sums meet.
template <int NC, int RW, int KB>
__global__ __launch_bounds__(512) void attn_bf16_beam_kernel(const bf16_t* __restrict__ ctx, const float* __restrict__ u, int64_t ldu, float* __restrict__ p_out,
                                                            float* __restrict__ o, int64_t ldo, int T, int k, bf16_t* __restrict__ ob, int64_t ldob) {
  constexpr int NW = 8, Hd = 512 * NC;
  extern __shared__ float sc[];                                 // [k][T] scores / probabilities
  __shared__ __attribute__((aligned(16))) float red[NW][Hd];
  const int img = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bf16_t* cb = ctx + (int64_t)img * T * Hd + lane * 8;
  float uu[KB][NC][8];
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const float* ub = u + (int64_t)(img * k + min(j, k - 1)) * ldu + lane * 8;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      const float4 u0 = *reinterpret_cast<const float4*>(ub + cc * 512), u1 = *reinterpret_cast<const float4*>(ub + cc * 512 + 4);
      uu[j][cc][0] = u0.x; uu[j][cc][1] = u0.y; uu[j][cc][2] = u0.z; uu[j][cc][3] = u0.w; uu[j][cc][4] = u1.x; uu[j][cc][5] = u1.y; uu[j][cc][6] = u1.z; uu[j][cc][7] = u1.w;
    }
  }
  bf16x8 c[RW][NC];
  auto load_rows = [&](int t0) {
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + wave + NW * i;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) {
        if (t < T) c[i][cc] = *reinterpret_cast<const bf16x8*>(cb + (int64_t)t * Hd + cc * 512);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) c[i][cc][e] = (bf16_t)0.f;
        }
      }
    }
  };
  // one pass (online softmax, as the streamed per-row kernel): running (m, l) and weighted sum per hypothesis and wave
  float m_run[KB], l_run[KB], acc[KB][NC][8];
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    m_run[j] = -INFINITY; l_run[j] = 0.f;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[j][cc][e] = 0.f;
  }
  for (int t0 = 0; t0 < T; t0 += NW * RW) {
    load_rows(t0);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int t = t0 + wave + NW * i;
      if (t < T) {
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          float sdot = 0.f;
#pragma unroll
          for (int cc = 0; cc < NC; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) sdot = fmaf((float)c[i][cc][e], uu[j][cc][e], sdot);
          sdot = wave_sum(sdot);
          if (lane == 0 && j < k) sc[(int64_t)j * T + t] = sdot;
          const float m_new = fmaxf(m_run[j], sdot), scale = __expf(m_run[j] - m_new), pe = __expf(sdot - m_new);
          l_run[j] = l_run[j] * scale + pe; m_run[j] = m_new;
#pragma unroll
          for (int cc = 0; cc < NC; ++cc)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][cc][e] = fmaf(pe, (float)c[i][cc][e], acc[j][cc][e] * scale);
        }
      }
    }
  }
  float* const ml = &red[0][0];                                 // [KB][NW][2] = (m, l) per hypothesis and wave
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < KB; ++j) { ml[(j * NW + wave) * 2] = m_run[j]; ml[(j * NW + wave) * 2 + 1] = l_run[j]; }
  }
  __syncthreads();
  float wsc[KB];
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    float M = -INFINITY, Lsum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) M = fmaxf(M, ml[(j * NW + q) * 2]);
#pragma unroll
    for (int q = 0; q < NW; ++q) Lsum += ml[(j * NW + q) * 2 + 1] * __expf(ml[(j * NW + q) * 2] - M);
    const float inv = 1.f / Lsum;
    wsc[j] = __expf(m_run[j] - M) * inv;
    if (j < k) for (int t = threadIdx.x; t < T; t += 64 * NW) p_out[(int64_t)(img * k + j) * T + t] = __expf(sc[(int64_t)j * T + t] - M) * inv;
  }
  __syncthreads();                                              // every thread has read (m, l) before the image is overwritten
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    if (j >= k) break;
    if (j > 0) __syncthreads();                                 // every thread is done with the image of hypothesis j - 1
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8]) = make_float4(acc[j][cc][0] * wsc[j], acc[j][cc][1] * wsc[j], acc[j][cc][2] * wsc[j], acc[j][cc][3] * wsc[j]);
      *reinterpret_cast<float4*>(&red[wave][cc * 512 + lane * 8 + 4]) = make_float4(acc[j][cc][4] * wsc[j], acc[j][cc][5] * wsc[j], acc[j][cc][6] * wsc[j], acc[j][cc][7] * wsc[j]);
    }
    __syncthreads();
    const int64_t row = (int64_t)img * k + j;
    for (int q = threadIdx.x; q < Hd; q += 64 * NW) {
      const float v = ((red[0][q] + red[1][q]) + (red[2][q] + red[3][q])) + ((red[4][q] + red[5][q]) + (red[6][q] + red[7][q]));
      o[row * ldo + q] = v;
      if (ob) ob[row * ldob + q] = (bf16_t)v;
    }
  }
}

constexpr int ATTN_NW = 16;
template <bool BWD>
static void attn_launch(hipStream_t s, const float* ctx, const float* u, int64_t ldu, const float* a_in, float* p_out, float* o,
                        int64_t ldo, int B, int T, int Hd, int ctx_div, bf16_t* ob, int64_t ldob, const bf16_t* ctxb, const float* cfwd = nullptr, int64_t ldcf = 0) {
  if (getenv("AOCR_ATTN_BWD_TWO_PASS")) cfwd = nullptr;          // A/B: the streamed backward kernel's two-pass form
#define AOCR_ATTN_BF16(NC, RW, STREAM) do {                                                                                   \
    if (64 * 1024 + (size_t)T * 4 > 64 * 1024)                                                                                  \
      (void)hipFuncSetAttribute((const void*)attn_bf16_kernel<BWD, NC, RW, STREAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)T * 4)); \
    hipLaunchKernelGGL((attn_bf16_kernel<BWD, NC, RW, STREAM>), dim3(B), dim3(1024), (size_t)T * sizeof(float), s, ctxb, u, ldu, a_in, p_out, o, \
                       ldo, T, ctx_div, ob, ldob, (const bf16_t*)nullptr, (float*)nullptr, (int64_t)0, cfwd, ldcf); } while (0)
  const bool bf_ok = ctxb && ldu % 4 == 0 && !getenv("AOCR_NO_ATTN_BF16");
  if constexpr (!BWD) {
    // beam decode over a long context: one workgroup per IMAGE (its k hypotheses share every context row that is loaded); k <= 5, rows = images x k
    if (bf_ok && ctx_div > 1 && ctx_div <= 5 && B % ctx_div == 0 && T > 64 && (Hd == 1024 || Hd == 512) && (size_t)ctx_div * T * 4 <= 96 * 1024 && !getenv("AOCR_NO_ATTN_BEAM_GROUP")) {
      const size_t dyn = (size_t)ctx_div * T * sizeof(float);
      if (Hd == 1024) {
        if (dyn + 32 * 1024 > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_bf16_beam_kernel<2, 4, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((attn_bf16_beam_kernel<2, 4, 5>), dim3(B / ctx_div), dim3(512), dyn, s, ctxb, u, ldu, p_out, o, ldo, T, ctx_div, ob, ldob);
      } else {
        if (dyn + 16 * 1024 > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_bf16_beam_kernel<1, 8, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL((attn_bf16_beam_kernel<1, 8, 5>), dim3(B / ctx_div), dim3(512), dyn, s, ctxb, u, ldu, p_out, o, ldo, T, ctx_div, ob, ldob);
      }
      return;
    }
  }
  if (T <= 64 && Hd == 512 && ctxb && ldu % 4 == 0)
    hipLaunchKernelGGL((attn_reg_h512_kernel<BWD, ATTN_NW>), dim3(B), dim3(64 * ATTN_NW), 0, s, ctxb, u, ldu, a_in, p_out, o, ldo, T, ctx_div, ob, ldob);
  else if (bf_ok && Hd == 512 && T <= 128) AOCR_ATTN_BF16(1, 8, false);
  else if (bf_ok && Hd == 512 && T <= 256) AOCR_ATTN_BF16(1, 16, false);
  else if (bf_ok && Hd == 512) AOCR_ATTN_BF16(1, 4, true);
  else if (bf_ok && Hd == 1024 && T <= 32 && !getenv("AOCR_ATTN_NW16"))
    hipLaunchKernelGGL((attn_bf16_kernel<BWD, 2, 4, false, 8>), dim3(B), dim3(512), (size_t)T * sizeof(float), s, ctxb, u, ldu, a_in, p_out, o, ldo, T, ctx_div, ob, ldob);
  else if (bf_ok && Hd == 1024 && T <= 64) AOCR_ATTN_BF16(2, 4, false);
  else if (bf_ok && Hd == 1024 && T <= 128) AOCR_ATTN_BF16(2, 8, false);
  else if (bf_ok && Hd == 1024) AOCR_ATTN_BF16(2, 4, true);
  else if (T <= 64 && Hd == 512)
    hipLaunchKernelGGL((attn_reg_kernel<2, BWD, float>), dim3(B), dim3(256), 0, s, ctx, u, ldu, a_in, p_out, o, ldo, T, ctx_div, ob, ldob);
  else if (T <= 64 && Hd == 256)
    hipLaunchKernelGGL((attn_reg_kernel<1, BWD, float>), dim3(B), dim3(256), 0, s, ctx, u, ldu, a_in, p_out, o, ldo, T, ctx_div, ob, ldob);
  else
    hipLaunchKernelGGL((attn_core_kernel<BWD>), dim3(B), dim3(256), (size_t)(T + 8) * sizeof(float), s, ctx, u, ldu, a_in, p_out, o, ldo,
                       T, Hd, ctx_div, ob, ldob);
}
void attention_forward(hipStream_t s, const float* ctx, const float* q, float* a, float* c, int64_t ldc, int B, int T, int Hd,
                       int ctx_div, bf16_t* cb, int64_t ldcb, const bf16_t* ctxb) {
  attn_launch<false>(s, ctx, q, (int64_t)Hd, nullptr, a, c, ldc, B, T, Hd, ctx_div, cb, ldcb, ctxb);
}
void attention_backward(hipStream_t s, const float* ctx, const float* q, const float* a, const float* dc, int64_t lddc,
                        float* ds, float* dq, int B, int T, int Hd, bf16_t* dqb, const bf16_t* ctxb, const float* cfwd, int64_t ldcf) {
  (void)q;
  attn_launch<true>(s, ctx, dc, lddc, a, ds, dq, (int64_t)Hd, B, T, Hd, 1, dqb, (int64_t)Hd, ctxb, cfwd, ldcf);
}

// Two-context forms (attn_bf16_kernel<..., DUAL>): Hd = 1024, T <= 64, bf16 context and its pre-multiplied copy ctxa = bf16(ctx W_a).  false: shape not taken.
bool attention_dual_ok(int T, int Hd, const bf16_t* ctxb, const bf16_t* ctxab) { return Hd == 1024 && T <= 64 && ctxb && ctxab && !getenv("AOCR_NO_CHAIN_CTXA"); }
void attention_forward_dual(hipStream_t s, const float* h_top, int64_t ldh, float* a, float* c, int64_t ldc, int B, int T, int ctx_div, bf16_t* cb, int64_t ldcb,
                            const bf16_t* ctxb, const bf16_t* ctxab) {
  if (T <= 32) hipLaunchKernelGGL((attn_bf16_kernel<false, 2, 4, false, 8, true>), dim3(B), dim3(512), (size_t)T * sizeof(float), s, ctxb, h_top, ldh, nullptr, a, c, ldc, T, ctx_div, cb, ldcb, ctxab, nullptr, (int64_t)0);
  else hipLaunchKernelGGL((attn_bf16_kernel<false, 2, 4, false, 16, true>), dim3(B), dim3(1024), (size_t)T * sizeof(float), s, ctxb, h_top, ldh, nullptr, a, c, ldc, T, ctx_div, cb, ldcb, ctxab, nullptr, (int64_t)0);
}
void attention_backward_dual(hipStream_t s, const float* a, const float* dc, int64_t lddc, float* ds, float* dq, bf16_t* dqb, float* dh_attn, int B, int T,
                             const bf16_t* ctxb, const bf16_t* ctxab) {
  if (T <= 32) hipLaunchKernelGGL((attn_bf16_kernel<true, 2, 4, false, 8, true>), dim3(B), dim3(512), (size_t)T * sizeof(float), s, ctxb, dc, lddc, a, ds, dq, (int64_t)1024, T, 1, dqb, (int64_t)1024, ctxab, dh_attn, (int64_t)1024);
  else hipLaunchKernelGGL((attn_bf16_kernel<true, 2, 4, false, 16, true>), dim3(B), dim3(1024), (size_t)T * sizeof(float), s, ctxb, dc, lddc, a, ds, dq, (int64_t)1024, T, 1, dqb, (int64_t)1024, ctxab, dh_attn, (int64_t)1024);
}

// d(ctx)[b,t,j] = sum_l a[l,b,t]*dc[l,b,j] + ds[l,b,t]*q[l,b,j]  (model.lua:652-653 accumulated over the decoder loop)
// One workgroup per (batch row, 64 columns): the L rows of d c and q for those columns and a 64-step chunk of a / d s are staged once in
// LDS (the first form had every thread walk all L rows of d c and q from L2: 1.6 GB of L2 traffic per call at C3 for 60 MB of data).
constexpr int DCTX_L = 32;                                      // decoder steps per pass (accumulators persist over the passes)
__global__ __launch_bounds__(256) void attn_dctx_kernel(const float* __restrict__ a_all, const float* __restrict__ ds_all,
                                                        const float* __restrict__ dc_all, int64_t lddc,
                                                        const float* __restrict__ q_all, float* __restrict__ dctx, int L, int B,
                                                        int T, int Hd) {
  __shared__ __attribute__((aligned(16))) float s_dc[DCTX_L][64], s_q[DCTX_L][64], s_a[DCTX_L][64], s_ds[DCTX_L][64];
  const int b = blockIdx.y, j0 = blockIdx.x * 64, tid = threadIdx.x;
  const int jq = tid & 15, ts = tid >> 4;                       // thread -> columns j0 + 4 jq .. + 3, steps ts, ts + 16, ts + 32, ts + 48 of a chunk
  for (int t0 = 0; t0 < T; t0 += 64) {
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l1 = L; l1 > 0; l1 -= DCTX_L) {                    // passes from the last decoder step down: the reference's t = L..1 order
      const int l0 = max(l1 - DCTX_L, 0), nl = l1 - l0;
      __syncthreads();
      for (int i = tid; i < nl * 16; i += 256) {
        const int l = i >> 4, c = (i & 15) * 4;
        const bool in = j0 + c < Hd;                            // (Hd is a multiple of 32: the last block of columns may be half empty)
        *reinterpret_cast<float4*>(&s_dc[l][c]) = in ? *reinterpret_cast<const float4*>(dc_all + ((int64_t)(l0 + l) * B + b) * lddc + j0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&s_q[l][c]) = in ? *reinterpret_cast<const float4*>(q_all + ((int64_t)(l0 + l) * B + b) * Hd + j0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int i = tid; i < nl * 64; i += 256) {
        const int l = i >> 6, t = t0 + (i & 63);
        s_a[l][i & 63] = t < T ? a_all[((int64_t)(l0 + l) * B + b) * T + t] : 0.f;
        s_ds[l][i & 63] = t < T ? ds_all[((int64_t)(l0 + l) * B + b) * T + t] : 0.f;
      }
      __syncthreads();
      for (int l = nl - 1; l >= 0; --l) {
        const float4 dc = *reinterpret_cast<const float4*>(&s_dc[l][4 * jq]), qv = *reinterpret_cast<const float4*>(&s_q[l][4 * jq]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float av = s_a[l][ts + 16 * k], dv = s_ds[l][ts + 16 * k];
          acc[k].x += av * dc.x + dv * qv.x; acc[k].y += av * dc.y + dv * qv.y; acc[k].z += av * dc.z + dv * qv.z; acc[k].w += av * dc.w + dv * qv.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = t0 + ts + 16 * k;
      if (t < T && j0 + 4 * jq < Hd) *reinterpret_cast<float4*>(dctx + ((int64_t)b * T + t) * Hd + j0 + 4 * jq) = acc[k];
    }
  }
}
void attention_dctx(hipStream_t s, const float* a_all, const float* ds_all, const float* dc_all, int64_t lddc, const float* q_all,
                    float* dctx, int L, int B, int T, int Hd) {
  hipLaunchKernelGGL(attn_dctx_kernel, dim3((Hd + 63) / 64, B), dim3(256), 0, s, a_all, ds_all, dc_all, lddc, q_all, dctx, L, B, T, Hd);
}

// =============================================================================================
// LogSoftMax + weighted NLL (+ gradient): output_projector.lua:6, criterion.lua:3-8, model.lua:644-648.
// row = t*Bt + b; target id = tgt[t*stride_t + b*stride_b] (1-based; PAD=1 has weight 0).
// =============================================================================================
__global__ __launch_bounds__(256) void lsm_nll_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ tgt,
                                                      int64_t st, int64_t sb, int Bt, float* __restrict__ logp,
                                                      float* __restrict__ dlogits, float* __restrict__ nll, int64_t rows, int V,
                                                      float scale) {
  int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float* x = logits + r * ld;
  float m = -INFINITY;
  for (int v = 0; v < V; ++v) m = fmaxf(m, x[v]);
  float sum = 0.f;
  for (int v = 0; v < V; ++v) sum += expf(x[v] - m);
  float lse = m + logf(sum);
  int64_t t = r / Bt, b = r - t * Bt;
  int y = tgt[t * st + b * sb] - 1;
  float wy = (y == 0) ? 0.f : 1.f;                              // criterion.lua:5: weights[PAD] = 0
  if (nll) nll[r] = -wy * (x[y] - lse);
  if (logp) for (int v = 0; v < V; ++v) logp[r * V + v] = x[v] - lse;
  if (dlogits) {
    float g = scale * wy;
    for (int v = 0; v < V; ++v) dlogits[r * ld + v] = g * (expf(x[v] - lse) - (v == y ? 1.f : 0.f));
    for (int v = V; v < ld; ++v) dlogits[r * ld + v] = 0.f;
  }
}
// V <= 64: one wave per row, lane = class (the one-thread-per-row form walks V three times serially: 16 us for 256 rows)
__global__ __launch_bounds__(256) void lsm_nll_wave_kernel(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ tgt,
                                                           int64_t st, int64_t sb, int Bt, float* __restrict__ logp,
                                                           float* __restrict__ dlogits, float* __restrict__ nll, int64_t rows, int V,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float xv = lane < V ? logits[r * ld + lane] : -INFINITY;
  const float m = wave_max(xv);
  const float sum = wave_sum(lane < V ? expf(xv - m) : 0.f);
  const float lse = m + logf(sum);
  const int64_t t = r / Bt, b = r - t * Bt;
  const int y = tgt[t * st + b * sb] - 1;
  const float wy = (y == 0) ? 0.f : 1.f;                        // criterion.lua:5: weights[PAD] = 0
  if (nll && lane == y) nll[r] = -wy * (xv - lse);
  if (logp && lane < V) logp[r * V + lane] = xv - lse;
  if (dlogits) {
    const float g = scale * wy;
    if (lane < V) dlogits[r * ld + lane] = g * (expf(xv - lse) - (lane == y ? 1.f : 0.f));
    for (int v = V + lane; v < ld; v += 64) dlogits[r * ld + v] = 0.f;
  }
}
void logsoftmax_nll(hipStream_t s, const float* logits, int64_t ld, const int32_t* tgt, int64_t st, int64_t sb, int Bt,
                    float* logp, float* dlogits, float* nll_rows, int64_t rows, int V, float grad_scale) {
  if (V <= 64) {
    hipLaunchKernelGGL(lsm_nll_wave_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, logits, ld, tgt, st, sb, Bt, logp, dlogits, nll_rows,
                       rows, V, grad_scale);
    return;
  }
  hipLaunchKernelGGL(lsm_nll_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, s, logits, ld, tgt, st, sb, Bt, logp, dlogits, nll_rows,
                     rows, V, grad_scale);
}

__global__ __launch_bounds__(256) void sum_scalar_kernel(const float* __restrict__ x, int64_t n, float* out) {
  __shared__ double sh[4];
  double s = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)x[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}
// Sum of the split-K slabs of a filter gradient (conv_wgrad_*_kernel with `part`): out[i] += sum_z part[z][i].  HBM-bound: every
// thread owns 4 consecutive floats, all `ks` 16-byte loads of a group of 8 slabs in flight before the first add.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ks, size_t n4, float* __restrict__ out) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f4* p = reinterpret_cast<const f4*>(part) + i;
  f4 acc = reinterpret_cast<const f4*>(out)[i];
  int z = 0;
  for (; z + 8 <= ks; z += 8) {
    f4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(p + (size_t)(z + j) * n4);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; z < ks; ++z) acc += __builtin_nontemporal_load(p + (size_t)z * n4);
  reinterpret_cast<f4*>(out)[i] = acc;
}
void splitk_reduce(hipStream_t s, const float* part, int ks, size_t n, float* out) {
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, ks, n4, out);
}
void sum_to_scalar(hipStream_t s, const float* x, int64_t n, float* out) {
  hipLaunchKernelGGL(sum_scalar_kernel, dim3(1), dim3(256), 0, s, x, n, out);
}
__global__ void gold_kernel(const float* __restrict__ nll, float* gold, int L, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int t = 0; t < L; ++t) s -= nll[(int64_t)t * B + b];     // model.lua:614-618 (PAD rows carry weight 0)
  gold[b] = s;
}
void gold_scores(hipStream_t s, const float* nll_rows, float* gold, int L, int B) {
  hipLaunchKernelGGL(gold_kernel, dim3(cdiv(B, 128)), dim3(128), 0, s, nll_rows, gold, L, B);
}

// out[n] += sum_r A[r][n]   (bias gradients).  grid = (column blocks of 64, row chunks); a workgroup is 16 row lanes x
// 16 column quads (dwordx4 loads, two rows in flight per lane), reduced through LDS; one atomic per column per workgroup.
__device__ __forceinline__ void colsum_body(const float* __restrict__ A, int64_t ld, int64_t rows, int N, float* out, float* out2, int bx, int by, int ny) {
  __shared__ float sh[16][65];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = bx * 64 + cq * 4;
  const int64_t per = (rows + ny - 1) / ny;
  const int64_t r0 = (int64_t)by * per, r1 = min(rows, r0 + per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), t = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec = (n + 3 < N) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  if (vec) {
    int64_t r = r0 + rl;
    for (; r + 7 * 16 < r1; r += 8 * 16) {                   // eight rows in flight per lane (two left the pass at ~1 TB/s on the decoder's 50 MB d z)
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(A + (r + 16 * u) * ld + n);
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; t.x += v[u + 1].x; t.y += v[u + 1].y; t.z += v[u + 1].z; t.w += v[u + 1].w;
      }
    }
    for (; r + 16 < r1; r += 32) {
      float4 a = *reinterpret_cast<const float4*>(A + r * ld + n), b = *reinterpret_cast<const float4*>(A + (r + 16) * ld + n);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    }
    for (; r < r1; r += 16) { float4 a = *reinterpret_cast<const float4*>(A + r * ld + n); s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
  } else {
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      if (n < N) s.x += A[r * ld + n];
      if (n + 1 < N) s.y += A[r * ld + n + 1];
      if (n + 2 < N) s.z += A[r * ld + n + 2];
      if (n + 3 < N) s.w += A[r * ld + n + 3];
    }
  }
  sh[rl][cq * 4] = s.x + t.x; sh[rl][cq * 4 + 1] = s.y + t.y; sh[rl][cq * 4 + 2] = s.z + t.z; sh[rl][cq * 4 + 3] = s.w + t.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = bx * 64 + threadIdx.x;
    if (c < N) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) v += sh[i][threadIdx.x];
      if (ny == 1) out[c] += v; else atomicAdd(&out[c], v);
      if (out2) { if (ny == 1) out2[c] += v; else atomicAdd(&out2[c], v); }      // second accumulator of the same sums (the two LSTM biases)
    }
  }
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int64_t ld, int64_t rows, int N, float* out, float* out2) {
  colsum_body(A, ld, rows, N, out, out2, blockIdx.x, blockIdx.y, gridDim.y);
}
__global__ __launch_bounds__(256) void colsum_jobs_kernel(ColsumJobs g) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) if (i < g.n && (int)blockIdx.x >= g.j[i].first) ji = i;
  const ColsumJob& J = g.j[ji];
  const int local = blockIdx.x - J.first;
  colsum_body(J.A, J.ld, J.rows, J.N, J.out, J.out2, local % J.nb, local / J.nb, J.chunks);
}
void colsum_accum(hipStream_t s, const float* A, int64_t ld, int64_t rows, int N, float* out, float* out2) {
  int nb = cdiv(N, 64);
  int chunks = (int)std::min<int64_t>(std::max<int64_t>(1, 2048 / nb), (rows + 511) / 512);
  hipLaunchKernelGGL(colsum_kernel, dim3(nb, chunks), dim3(256), 0, s, A, ld, rows, N, out, out2);
}
void colsum_defer(ColsumJobs& g, const float* A, int64_t ld, int64_t rows, int N, float* out, float* out2) {
  if (g.n >= 8) return;                                         // (callers flush before the table is full)
  ColsumJob& J = g.j[g.n];
  J.A = A; J.ld = ld; J.rows = rows; J.N = N; J.out = out; J.out2 = out2;
  J.nb = cdiv(N, 64);
  J.chunks = (int)std::min<int64_t>(std::max<int64_t>(1, 2048 / J.nb), (rows + 511) / 512);
  if (J.chunks < 1) J.chunks = 1;
  J.first = g.total; g.total += J.nb * J.chunks; ++g.n;
}
void colsum_flush(hipStream_t s, ColsumJobs& g) {
  if (g.n > 0) hipLaunchKernelGGL(colsum_jobs_kernel, dim3(g.total), dim3(256), 0, s, g);
  g.n = 0; g.total = 0;
}

// element-wise helpers of the module-level surface (AOCR_PW_* of include/aocr.h): 16-byte accesses on the aligned body, HBM-bound
__global__ __launch_bounds__(256) void pointwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n, int vec) {
  auto f = [op](float x, float z) { return op == 0 ? x + z : op == 1 ? x * (1.f - z * z) : op == 2 ? (z > 0.f ? x : 0.f) : fmaxf(x, 0.f); };
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (vec) {
    if (i * 4 + 3 < n) {
      const float4 x = reinterpret_cast<const float4*>(a)[i]; const float4 z = b ? reinterpret_cast<const float4*>(b)[i] : x;
      reinterpret_cast<float4*>(y)[i] = make_float4(f(x.x, z.x), f(x.y, z.y), f(x.z, z.z), f(x.w, z.w));
    } else for (int64_t j = i * 4; j < n; ++j) y[j] = f(a[j], b ? b[j] : a[j]);
  } else if (i < n) y[i] = f(a[i], b ? b[i] : a[i]);
}
void pointwise(hipStream_t s, int op, const float* a, const float* b, float* y, int64_t n) {
  if (n <= 0) return;
  const int vec = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  const int64_t items = vec ? (n + 3) / 4 : n;
  hipLaunchKernelGGL(pointwise_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, op, a, b, y, n, vec);
}

// nn.LookupTable forward / accGradParameters (LSTM.lua:55-56)
__global__ void emb_gather_kernel(const float* __restrict__ table, const int32_t* __restrict__ tok, int64_t st, int64_t sb,
                                  float* out, int L, int B, int E) {
  int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (int64_t)L * B * E) return;
  int e = (int)(id % E); int64_t r = id / E; int64_t t = r / B, b = r - t * B;
  out[id] = table[(int64_t)(tok[t * st + b * sb] - 1) * E + e];
}
void embedding_gather(hipStream_t s, const float* table, const int32_t* tok, int64_t st, int64_t sb, float* out, int L, int B, int E) {
  int64_t n = (int64_t)L * B * E;
  hipLaunchKernelGGL(emb_gather_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, table, tok, st, sb, out, L, B, E);
}
__global__ __launch_bounds__(256) void emb_scatter_kernel(const float* __restrict__ demb, const int32_t* __restrict__ tok, int64_t st,
                                                          int64_t sb, float* dtable, int L, int B, int E) {
  extern __shared__ float part[];                               // [row lanes][E]; one workgroup per (vocabulary row, row slice)
  const int v = blockIdx.x;
  const int per = (L * B + gridDim.y - 1) / gridDim.y;          // the rows are split over gridDim.y slices (one atomic per slice)
  const int rbeg = blockIdx.y * per, rows = min(L * B, rbeg + per);
  const int nrl = 256 / E;                                      // row lanes (E <= 256)
  const int e = threadIdx.x % E, rl = threadIdx.x / E;
  float s = 0.f;
  if (rl < nrl) {
#pragma unroll 8
    for (int r = rbeg + rl; r < rows; r += nrl) {               // unconditional loads: 8 independent rows in flight
      int t = r / B, b = r - t * B;
      const float d = demb[(int64_t)r * E + e];
      s += (tok[t * st + b * sb] - 1 == v) ? d : 0.f;
    }
  }
  if (rl < nrl) part[rl * E + e] = s;
  __syncthreads();
  if ((int)threadIdx.x < E) {
    float t = 0.f;
    for (int i = 0; i < nrl; ++i) t += part[i * E + threadIdx.x];
    if (gridDim.y == 1) dtable[(int64_t)v * E + threadIdx.x] += t;
    else atomicAdd(&dtable[(int64_t)v * E + threadIdx.x], t);
  }
}
void embedding_scatter_accum(hipStream_t s, const float* demb, const int32_t* tok, int64_t st, int64_t sb, float* dtable, int L,
                             int B, int E, int V) {
  const int slices = (int64_t)L * B >= 2048 ? 8 : 1;           // 39 workgroups walking 6144 rows each took 52 us at C3
  hipLaunchKernelGGL(emb_scatter_kernel, dim3(V, slices), dim3(256), (size_t)256 * sizeof(float), s, demb, tok, st, sb, dtable, L, B, E);
}

// Round 4: the embedding side of the first decoder layer without the (rows, E) detour.  Every row's embedding is one of V table rows, so
//   d lookup[v] = (sum of d z over the rows whose token is v) . W_i2h[:, :E],   d W_i2h[:, :E] = (those sums)^T . lookup,   d b = their total:
// one pass over d z that sums rows BY TOKEN (S [V][ncols]) replaces the K = 4 Hd product into d emb, its scatter, the (d z, emb) weight-gradient
// problem and the bias column-sum pass over the same 50 MB (LSTM.lua:55-56 nn.LookupTable accGradParameters; model.lua:643-661).
// The rows are first grouped by token (one small workgroup: counting sort of the L B tokens into an index list + a list of work items of at
// most SEG_CHUNK rows of ONE token), then every work item sums its rows in registers -- whole 4 KB row pieces, eight rows in flight -- and adds
// its float4 to S with one atomic per component.  (First version: every workgroup kept [V][256] LDS accumulators over a slice of rows and
// issued one global atomic per (token seen, column): 3.9 M device-scope atomics, 49 us for the 50 MB at C3.)
constexpr int SEG_CHUNK = 48;
__global__ __launch_bounds__(1024) void token_sort_kernel(const int32_t* __restrict__ tok, int64_t st, int64_t sb, int rows, int B, int V,
                                                          int* __restrict__ order, int* __restrict__ items /* [n][3] = token, begin, end */, int* __restrict__ nitems) {
  __shared__ int cnt[64], pos[64], start[64], ifirst[64];
  const int tid = threadIdx.x;
  if (tid < 64) cnt[tid] = 0;
  __syncthreads();
  int mine[8];                                                  // this thread's rows (rows <= 8192 per call; more rows loop again below)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int r = tid + 1024 * u; mine[u] = -1;
    if (r < rows) { const int t = r / B, b = r - t * B; mine[u] = min(max(tok[t * st + b * sb], 1), V) - 1; atomicAdd(&cnt[mine[u]], 1); }
  }
  for (int r = tid + 8192; r < rows; r += 1024) { const int t = r / B, b = r - t * B; atomicAdd(&cnt[min(max(tok[t * st + b * sb], 1), V) - 1], 1); }
  __syncthreads();
  if (tid == 0) {
    int a = 0, n = 0;
    for (int v = 0; v < V; ++v) { start[v] = a; pos[v] = a; ifirst[v] = n; a += cnt[v]; n += (cnt[v] + SEG_CHUNK - 1) / SEG_CHUNK; }
    *nitems = n;
  }
  __syncthreads();
  if (tid < V) {                                                // every token emits its own work items
    int n = ifirst[tid];
    for (int b0 = start[tid]; b0 < start[tid] + cnt[tid]; b0 += SEG_CHUNK, ++n) { items[3 * n] = tid; items[3 * n + 1] = b0; items[3 * n + 2] = min(start[tid] + cnt[tid], b0 + SEG_CHUNK); }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) if (mine[u] >= 0) order[atomicAdd(&pos[mine[u]], 1)] = tid + 1024 * u;
  for (int r = tid + 8192; r < rows; r += 1024) { const int t = r / B, b = r - t * B; order[atomicAdd(&pos[min(max(tok[t * st + b * sb], 1), V) - 1], 1)] = r; }
}
__global__ __launch_bounds__(256) void segsum_sorted_kernel(const float* __restrict__ dz, int64_t ld, const int* __restrict__ order, const int* __restrict__ items,
                                                            const int* __restrict__ nitems, int ncols, float* __restrict__ S) {
  __shared__ int idx[SEG_CHUNK];
  if ((int)blockIdx.x >= *nitems) return;
  const int v = items[3 * blockIdx.x], r0 = items[3 * blockIdx.x + 1], n = items[3 * blockIdx.x + 2] - r0;
  if ((int)threadIdx.x < n) idx[threadIdx.x] = order[r0 + threadIdx.x];      // the item's row list once, through LDS: the row loads below do not wait for an index load each
  __syncthreads();
  const int col = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (col >= ncols) return;                                      // (ncols % 4 == 0)
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
  int r = 0;
  for (; r + 16 <= n; r += 16) {                                 // sixteen rows in flight
    float4 x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = *reinterpret_cast<const float4*>(dz + (int64_t)idx[r + u] * ld + col);
#pragma unroll
    for (int u = 0; u < 16; u += 2) { a.x += x[u].x; a.y += x[u].y; a.z += x[u].z; a.w += x[u].w; c.x += x[u + 1].x; c.y += x[u + 1].y; c.z += x[u + 1].z; c.w += x[u + 1].w; }
  }
  for (; r < n; ++r) { const float4 x = *reinterpret_cast<const float4*>(dz + (int64_t)idx[r] * ld + col); a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
  float* o = S + (int64_t)v * ncols + col;
  atomicAdd(o, a.x + c.x); atomicAdd(o + 1, a.y + c.y); atomicAdd(o + 2, a.z + c.z); atomicAdd(o + 3, a.w + c.w);
}
// what hangs off S [V][ncols] (ncols = 4 Hd): the bias gradients (column totals), d lookup [V][E] += S W[:, :E] (W [ncols][ldw]) and
// d W[:, :E] += S^T lookup -- V-sized products in exact fp32
__global__ __launch_bounds__(256) void segsum_bias_kernel(const float* __restrict__ S, int V, int ncols, float* __restrict__ db1, float* __restrict__ db2) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncols) return;
  float s = 0.f;
  for (int v0 = 0; v0 < V; v0 += 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = v0 + u < V ? S[(int64_t)(v0 + u) * ncols + col] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u];
  }
  db1[col] += s; if (db2) db2[col] += s;
}
__global__ __launch_bounds__(256) void segsum_dlookup_kernel(const float* __restrict__ S, int V, int ncols, const float* __restrict__ W, int64_t ldw, int E, float* __restrict__ dlookup) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per output (v, e)
  if (o >= V * E) return;
  const int v = o / E, e = o - v * E;
  float acc = 0.f;
  for (int j0 = lane; j0 < ncols; j0 += 64 * 8) {
    float sv[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int j = j0 + 64 * u; const bool ok = j < ncols; sv[u] = ok ? S[(int64_t)v * ncols + j] : 0.f; w[u] = ok ? W[(int64_t)j * ldw + e] : 0.f; }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(sv[u], w[u], acc);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
  if (lane == 0) dlookup[o] += acc;
}
__global__ __launch_bounds__(256) void segsum_dw_kernel(const float* __restrict__ S, int V, int ncols, const float* __restrict__ lookup, int E, float* __restrict__ dW, int64_t ldw) {
  extern __shared__ float lk[];                                 // [V][E]
  for (int i = threadIdx.x; i < V * E; i += 256) lk[i] = lookup[i];
  __syncthreads();
  const int o = blockIdx.x * 256 + threadIdx.x;                 // one thread per output (j, e)
  if (o >= ncols * E) return;
  const int j = o / E, e = o - j * E;
  float acc = 0.f;
  for (int v0 = 0; v0 < V; v0 += 8) {
    float sv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) sv[u] = v0 + u < V ? S[(int64_t)(v0 + u) * ncols + j] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) if (v0 + u < V) acc = fmaf(sv[u], lk[(v0 + u) * E + e], acc);
  }
  dW[(int64_t)j * ldw + e] += acc;
}
bool segsum_supported(int ncols, int V, int E) { return ncols % 4 == 0 && V <= 64 && V * E <= 8192; }
size_t segsum_index_ints(int rows, int V) { return (size_t)rows + 3 * ((size_t)rows / SEG_CHUNK + V + 8) + 8; }
void segsum_by_token(hipStream_t s, const float* dz, int64_t ld, const int32_t* tok, int64_t st, int64_t sb, int L, int B, int ncols, int V, float* S, int* index,
                     float* db1, float* db2, const float* W, int64_t ldw, const float* lookup, int E, float* dlookup, float* dW) {
  const int rows = L * B;
  int* order = index; int* nitems = index + rows; int* items = nitems + 8;
  const int max_items = rows / SEG_CHUNK + V + 1;
  (void)hipMemsetAsync(S, 0, (size_t)V * ncols * sizeof(float), s);
  hipLaunchKernelGGL(token_sort_kernel, dim3(1), dim3(1024), 0, s, tok, st, sb, rows, B, V, order, items, nitems);
  hipLaunchKernelGGL(segsum_sorted_kernel, dim3(max_items, cdiv(ncols, 1024)), dim3(256), 0, s, dz, ld, order, items, nitems, ncols, S);
  hipLaunchKernelGGL(segsum_bias_kernel, dim3(cdiv(ncols, 256)), dim3(256), 0, s, S, V, ncols, db1, db2);
  hipLaunchKernelGGL(segsum_dlookup_kernel, dim3(cdiv(V * E, 4)), dim3(256), 0, s, S, V, ncols, W, ldw, E, dlookup);
  hipLaunchKernelGGL(segsum_dw_kernel, dim3(cdiv((int64_t)ncols * E, 256)), dim3(256), (size_t)V * E * sizeof(float), s, S, V, ncols, lookup, E, dW, ldw);
}

__global__ __launch_bounds__(256) void dpre_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                   const float* __restrict__ out, float* __restrict__ dpre, int64_t n, bf16_t* __restrict__ dpreb, DropSpec drop) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float g = g1[i] + (g2 ? g2[i] : 0.f);
  float o = out[i];
  if (drop.on()) { const float mk = drop.mask(i); o = mk != 0.f ? o / mk : 0.f; g *= mk; }      // out holds the masked value: tanh' needs the unmasked one
  const float v = g * (1.f - o * o);
  dpre[i] = v;
  if (dpreb) dpreb[i] = (bf16_t)v;
}
// debugging taps: the pooling arg-max maps as floats
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}
void u8_to_f32(hipStream_t s, const uint8_t* src, float* dst, int64_t n) {
  hipLaunchKernelGGL(u8_to_f32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, n);
}
void dpre_tanh(hipStream_t s, const float* g1, const float* g2, const float* out, float* dpre, int64_t n, bf16_t* dpreb, const DropSpec* drop) {
  hipLaunchKernelGGL(dpre_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, g1, g2, out, dpre, n, dpreb, drop ? *drop : DropSpec{});
}
// dropout of a whole buffer (encoder layers above the first): dst = mask * src (+ bf16 copy); in place for the backward scaling
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dstb, int64_t n, DropSpec drop) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = src[i] * drop.mask(i);
  if (dst) dst[i] = v;
  if (dstb) dstb[i] = (bf16_t)v;
}
void dropout_apply(hipStream_t s, const float* src, float* dst, bf16_t* dstb, int64_t n, const DropSpec& drop) {
  hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, dstb, n, drop);
}
// the same from a bf16 source (the encoder cluster kernels keep h as bf16 only)
__global__ __launch_bounds__(256) void dropout_apply_b_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dstb, int64_t n, DropSpec drop) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = (float)src[i] * drop.mask(i);
  if (dst) dst[i] = v;
  if (dstb) dstb[i] = (bf16_t)v;
}
void dropout_apply_b(hipStream_t s, const bf16_t* src, float* dst, bf16_t* dstb, int64_t n, const DropSpec& drop) {
  hipLaunchKernelGGL(dropout_apply_b_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, dst, dstb, n, drop);
}
// several small regions zeroed by ONE launch (each hipMemsetAsync is its own ~5 us dispatch on the step's critical path)
__global__ __launch_bounds__(256) void zero_many_kernel(ZeroList z) {
  unsigned char* p = (unsigned char*)z.p[blockIdx.y];
  const size_t n = z.bytes[blockIdx.y], n16 = n >> 4;
  u32x4v* p16 = reinterpret_cast<u32x4v*>(p);            // regions are 16-byte aligned (workspace carve)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p16[i] = u32x4v{0u, 0u, 0u, 0u};
  if (blockIdx.x == 0) for (size_t i = (n16 << 4) + threadIdx.x; i < n; i += 256) p[i] = 0;
}
void zero_many(hipStream_t s, const ZeroList& z) {
  if (z.n <= 0) return;
  hipLaunchKernelGGL(zero_many_kernel, dim3(32, z.n), dim3(256), 0, s, z);
}
__global__ __launch_bounds__(256) void copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd,
                                                     int rows, int cols) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  int c = (int)(i % cols); int64_t r = i / cols;
  dst[r * ldd + c] = src[r * lds + c];
}
__global__ __launch_bounds__(256) void copy2d_bf16_kernel(const float* __restrict__ src, int64_t lds, bf16_t* __restrict__ dst, int64_t ldd,
                                                          int rows, int cols) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  int c = (int)(i % cols); int64_t r = i / cols;
  dst[r * ldd + c] = (bf16_t)src[r * lds + c];
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = (float)src[i];
}
void bf16_to_f32(hipStream_t s, const bf16_t* src, float* dst, int64_t n) {
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((int)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, s, src, dst, n);
}
__global__ __launch_bounds__(256) void copy2d_bf16x4_kernel(const float* __restrict__ src, int64_t lds, bf16_t* __restrict__ dst, int64_t ldd,
                                                            int rows, int cols4) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols4) return;
  const int c = (int)(i % cols4) * 4; const int64_t r = i / cols4;
  const float4 v = *reinterpret_cast<const float4*>(src + r * lds + c);
  bf16x4 hb; hb[0] = (bf16_t)v.x; hb[1] = (bf16_t)v.y; hb[2] = (bf16_t)v.z; hb[3] = (bf16_t)v.w;
  *reinterpret_cast<bf16x4*>(dst + r * ldd + c) = hb;
}
void copy2d_bf16(hipStream_t s, const float* src, int64_t lds, bf16_t* dst, int64_t ldd, int rows, int cols) {
  if (((cols | lds | ldd) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0) {
    hipLaunchKernelGGL(copy2d_bf16x4_kernel, dim3(cdiv((int64_t)rows * (cols / 4), 256)), dim3(256), 0, s, src, lds, dst, ldd, rows, cols / 4);
    return;
  }
  hipLaunchKernelGGL(copy2d_bf16_kernel, dim3(cdiv((int64_t)rows * cols, 256)), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
}
// Initial decoder state in ONE launch (model.lua:541-552): c_1(0) = [c_fw(T) ; c_bw(1)] (and h_1(0) likewise unless quirk S5 zeroes it), every other
// state, the first input feed and their bf16 shadows zero.  Was a zero-list launch + 2-4 strided copies, 3-5 dependent ~6 us dispatches per call.
__global__ __launch_bounds__(256) void dec_init_kernel(DecInitArgs a) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)a.B * a.Hd) return;
  const int r = (int)(idx / a.Hd), j = (int)(idx - (int64_t)r * a.Hd);
  const bool fw = j < a.He, in = j < 2 * a.He;
  const int64_t e = (int64_t)r * a.He + (fw ? j : j - a.He);
  const float c = in ? (fw ? a.cfw[e] : a.cbw[e]) : 0.f;
  const float h = (in && a.copy_h) ? (fw ? a.hfw[e] : a.hbw[e]) : 0.f;
  for (int l = 0; l < a.Ld; ++l) {
    a.c0[l][idx] = l == 0 ? c : 0.f;
    const float hv = l == 0 ? h : 0.f;
    a.h0[l][idx] = hv;
    if (a.hb[l]) a.hb[l][idx] = (bf16_t)hv;
  }
  if (a.feed0) a.feed0[idx] = 0.f;
  if (a.outb) a.outb[idx] = (bf16_t)0.f;
}
void dec_init(hipStream_t s, const DecInitArgs& a) {
  hipLaunchKernelGGL(dec_init_kernel, dim3(cdiv((int64_t)a.B * a.Hd, 256)), dim3(256), 0, s, a);
}
// two strided copies of the same shape in one launch (blockIdx.y picks the pair): the two halves of the decoder's initial-cell gradient -> the two encoder directions
__global__ __launch_bounds__(256) void copy2d_pair_kernel(const float* __restrict__ s0, const float* __restrict__ s1, int64_t lds, float* __restrict__ d0, float* __restrict__ d1, int64_t ldd, int rows, int cols) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  const float* src = blockIdx.y ? s1 : s0; float* dst = blockIdx.y ? d1 : d0;
  dst[(int64_t)r * ldd + c] = src[(int64_t)r * lds + c];
}
void copy2d_pair(hipStream_t s, const float* s0, const float* s1, int64_t lds, float* d0, float* d1, int64_t ldd, int rows, int cols) {
  hipLaunchKernelGGL(copy2d_pair_kernel, dim3(cdiv((int64_t)rows * cols, 256), 2), dim3(256), 0, s, s0, s1, lds, d0, d1, ldd, rows, cols);
}
void copy2d(hipStream_t s, const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int cols) {
  hipLaunchKernelGGL(copy2d_kernel, dim3(cdiv((int64_t)rows * cols, 256)), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
}

// =============================================================================================
// optim.sgd_list (optim_sgd.lua:38-95): per-group L2 norms (fp64 accumulation), clip, update.  No host sync.
// scratch: double part[5][2][SGD_BLOCKS]; float scale[5].
// =============================================================================================
static const int SGD_BLOCKS = 256;
size_t sgd_scratch_bytes() { return (size_t)5 * 2 * SGD_BLOCKS * sizeof(double) + 8 * sizeof(float); }
struct GroupOff { int64_t o[6]; };

__global__ __launch_bounds__(256) void sgd_sumsq_kernel(const float* __restrict__ p, const float* __restrict__ g, GroupOff go,
                                                        double* __restrict__ part) {
  __shared__ double sh[2][4];
  const int grp = blockIdx.y;
  const int64_t beg = go.o[grp], end = go.o[grp + 1];
  double sp = 0, sg = 0;
  const int64_t b4 = min(end, (beg + 3) & ~(int64_t)3), e4 = max(b4, end & ~(int64_t)3);      // 16-byte aligned body, scalar edges
  if (blockIdx.x == 0 && threadIdx.x < 8) {                     // at most 3 + 3 edge elements
    const int64_t i = threadIdx.x < 4 ? beg + threadIdx.x : e4 + (threadIdx.x - 4);
    if ((threadIdx.x < 4 && i < b4) || (threadIdx.x >= 4 && i < end)) { double a = p[i], b = g[i]; sp += a * a; sg += b * b; }
  }
  for (int64_t i = b4 + ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < e4; i += (int64_t)gridDim.x * 1024) {
    const float4 a = *reinterpret_cast<const float4*>(p + i), b = *reinterpret_cast<const float4*>(g + i);
    sp += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    sg += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
  }
  sp = wave_sum_d(sp); sg = wave_sum_d(sg);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = sp; sh[1][threadIdx.x >> 6] = sg; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((int64_t)grp * 2 + 0) * SGD_BLOCKS + blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    part[((int64_t)grp * 2 + 1) * SGD_BLOCKS + blockIdx.x] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}
// The time-out word of the whole-sequence kernels (cl_err, include/aocr.h: aocr_cluster_status): word 0 = code of the step in flight, CL_ERR_LATCH = the
// decision THIS optimizer call acts on, CL_ERR_STICKY = the last code the host has not read yet.  The optimizer consumes word 0 (round 6):
// only the step that timed out is skipped, whenever the host happens to poll.
__device__ __forceinline__ void cl_err_latch(int* err) {
  const int c = err[0];
  err[CL_ERR_LATCH] = c;
  if (c != 0) { err[CL_ERR_STICKY] = c; err[0] = 0; }
}
__global__ void cl_err_latch_kernel(int* err) { cl_err_latch(err); }
// start of a training step: snapshot of the BatchNorm running statistics (an optimizer call that skips its update restores it) and carry of a code no
// optimizer call has consumed -- a decode call's -- into the sticky word, so that it cannot cancel this step's update
__global__ __launch_bounds__(256) void step_snapshot_kernel(const float* __restrict__ bn_state, float* __restrict__ bn_snap, int n, int* err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) bn_snap[i] = bn_state[i];
  if (i == 0 && err && err[0] != 0) { err[CL_ERR_STICKY] = err[0]; err[0] = 0; }
}
void step_snapshot(hipStream_t s, const float* bn_state, float* bn_snap, int n, int* err) {
  hipLaunchKernelGGL(step_snapshot_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, bn_state, bn_snap, n, err);
}
__global__ void sgd_scale_kernel(const double* __restrict__ part, float clip, float* scale, float* norms, int* err) {
  const int grp = blockIdx.x, lane = threadIdx.x;               // one wave per group
  if (err && grp == 0 && lane == 0) cl_err_latch(err);          // (this launch sits between the last kernel that can raise the code and the update that reads the latch)
  double sp = 0, sg = 0;
  for (int i = lane; i < SGD_BLOCKS; i += 64) { sp += part[((int64_t)grp * 2) * SGD_BLOCKS + i]; sg += part[((int64_t)grp * 2 + 1) * SGD_BLOCKS + i]; }
  sp = wave_sum_d(sp); sg = wave_sum_d(sg);
  if (lane != 0) return;
  double pn = sqrt(sp), gn = sqrt(sg);
  scale[grp] = (gn > (double)clip) ? (float)((double)clip / gn) : 1.f;       // optim_sgd.lua:50-52
  if (norms) { norms[grp * 2] = (float)pn; norms[grp * 2 + 1] = (float)gn; }
}
__global__ __launch_bounds__(256) void sgd_update_kernel(float* __restrict__ p, const float* __restrict__ g, GroupOff go,
                                                         const float* __restrict__ scale, float lr, const int* __restrict__ skip,
                                                         float* __restrict__ bn_state, const float* __restrict__ bn_snap, int bn_n) {
  if (skip && *skip != 0) {            // a whole-sequence kernel of this step gave up waiting for its group (aocr_cluster_status): its gradients are invalid, keep the parameters
    // ... and take back the step's move of the BatchNorm running statistics (snapshot from the start of the step): the host repeats the
    // batch, and the repeat moves them once -- whenever the host happens to poll the status
    if (bn_state && blockIdx.y == 0)
      for (int i = blockIdx.x * 256 + threadIdx.x; i < bn_n; i += gridDim.x * 256) bn_state[i] = bn_snap[i];
    return;
  }
  const int grp = blockIdx.y;
  const int64_t beg = go.o[grp], end = go.o[grp + 1];
  const float sc = scale[grp];
  const int64_t b4 = min(end, (beg + 3) & ~(int64_t)3), e4 = max(b4, end & ~(int64_t)3);
  if (blockIdx.x == 0 && threadIdx.x < 8) {                     // at most 3 + 3 edge elements
    const int64_t i = threadIdx.x < 4 ? beg + threadIdx.x : e4 + (threadIdx.x - 4);
    if ((threadIdx.x < 4 && i < b4) || (threadIdx.x >= 4 && i < end)) p[i] = p[i] - lr * (g[i] * sc);     // optim_sgd.lua:52,90
  }
  for (int64_t i = b4 + ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < e4; i += (int64_t)gridDim.x * 1024) {
    float4 a = *reinterpret_cast<const float4*>(p + i); const float4 b = *reinterpret_cast<const float4*>(g + i);
    a.x = a.x - lr * (b.x * sc); a.y = a.y - lr * (b.y * sc); a.z = a.z - lr * (b.z * sc); a.w = a.w - lr * (b.w * sc);
    *reinterpret_cast<float4*>(p + i) = a;
  }
}
void sgd_clip_update(hipStream_t s, float* params, float* grads, const int64_t* group_off, float lr, float clip, float* norms_out,
                     void* scratch, int* err, float* bn_state, const float* bn_snap, int bn_n) {
  const int* skip = err ? err + CL_ERR_LATCH : nullptr;
  GroupOff go; for (int i = 0; i < 6; ++i) go.o[i] = group_off[i];
  double* part = (double*)scratch; float* scale = (float*)(part + 5 * 2 * SGD_BLOCKS);
  hipLaunchKernelGGL(sgd_sumsq_kernel, dim3(SGD_BLOCKS, 5), dim3(256), 0, s, params, grads, go, part);
  hipLaunchKernelGGL(sgd_scale_kernel, dim3(5), dim3(64), 0, s, part, clip, scale, norms_out, err);
  hipLaunchKernelGGL(sgd_update_kernel, dim3(512, 5), dim3(256), 0, s, params, grads, go, scale, lr, skip, bn_state, bn_snap, bn_n);
}

// optim.adadelta_list, src/optim/optim_adadelta.lua:19-62, fused into one pass over the flat vectors (the reference walks the 5
// groups with 9 tensor ops each).  var = rho*var + (1-rho) g^2; delta = sqrt(acc+eps)/sqrt(var+eps) * g; x -= delta;
// acc = rho*acc + (1-rho) delta^2.  Weight decay is what line 37 means (g += wd*x; the line itself indexes the table of
// gradients and would raise).  HBM-bound: 4 reads + 3 writes of 4 bytes per parameter.
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ x, float* __restrict__ g, float* __restrict__ var,
                                                       float* __restrict__ acc, int64_t n, float rho, float eps, float wd, const int* __restrict__ skip,
                                                       float* __restrict__ bn_state, const float* __restrict__ bn_snap, int bn_n) {
  if (skip && *skip != 0) {            // see sgd_update_kernel
    if (bn_state)
      for (int i = blockIdx.x * 256 + threadIdx.x; i < bn_n; i += gridDim.x * 256) bn_state[i] = bn_snap[i];
    return;
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float xv = x[i];
    float gv = g[i];
    if (wd != 0.f) { gv = fmaf(wd, xv, gv); g[i] = gv; }
    const float v = var[i] * rho + (1.f - rho) * gv * gv;                        // :48
    const float d = sqrtf(acc[i] + eps) / sqrtf(v + eps) * gv;                   // :49-50
    x[i] = xv - d;                                                               // :51
    acc[i] = acc[i] * rho + (1.f - rho) * d * d;                                 // :52
    var[i] = v;
  }
}
void adadelta_update(hipStream_t s, float* params, float* grads, float* var, float* acc, int64_t n, float rho, float eps, float wd, int* err,
                     float* bn_state, const float* bn_snap, int bn_n) {
  const int* skip = err ? err + CL_ERR_LATCH : nullptr;
  if (err) hipLaunchKernelGGL(cl_err_latch_kernel, dim3(1), dim3(1), 0, s, err);
  hipLaunchKernelGGL(adadelta_kernel, dim3(2048), dim3(256), 0, s, params, grads, var, acc, n, rho, eps, wd, skip, bn_state, bn_snap, bn_n);
}

// =============================================================================================
// beam search bookkeeping, model.lua:399-404, 446-458, 516-535, 573-585.
// =============================================================================================
// Dictionary constraint (model.lua:380-387,405-445,460-513 over the trie of utils.lua:177-218).  The trie is flat: node n has a
// child for vocab id v (1-based) iff bit v-1 of mask[n] is set; its children are child[base[n] ..] in ascending v.  Node 0 is the
// start symbol's node (trie[2]).  A candidate (beam, v) is admissible iff the beam's node has that child, or -- after the first
// step -- v is PAD (model.lua:469); PAD keeps the node, any other token moves to the child (:499-506).
__device__ __forceinline__ bool trie_ok(const TrieView& tv, int node, int v0, bool first) {
  return (!first && v0 == 0) || ((tv.mask[node] >> v0) & 1ull);
}
__device__ __forceinline__ int trie_next(const TrieView& tv, int node, int v0, bool first) {
  if (!first && v0 == 0) return node;
  const unsigned long long mk = tv.mask[node];
  if (!((mk >> v0) & 1ull)) return node;                        // only reachable when the trie admits nothing at all
  return tv.child[tv.base[node] + __popcll(mk & ((1ull << v0) - 1ull))];
}
// one wave per batch row; candidates c = beam*V + v (v 0-based).  Selection: descending score, ties -> lowest index.
__global__ __launch_bounds__(64) void beam_select_kernel(const float* __restrict__ logp, const int32_t* __restrict__ prev_tok,
                                                         float* __restrict__ beam_scores, int32_t* __restrict__ tokens,
                                                         int32_t* __restrict__ parents, int kin, int kout, int V,
                                                         const float* __restrict__ logits, int64_t ldl, TrieView tv) {
  extern __shared__ float cand[];                               // kin*V
  const int b = blockIdx.x, lane = threadIdx.x;
  const bool first = prev_tok == nullptr;
  const int n = kin * V;
  if (logits) {                                                 // fused LogSoftMax (V <= 64: lane = class), output_projector.lua:6
    for (int beam = 0; beam < kin; ++beam) {
      const int row = b * kin + beam;
      const float xv = lane < V ? logits[(int64_t)row * ldl + lane] : -INFINITY;
      const float mx = wave_max(xv);
      const float sum = wave_sum(lane < V ? expf(xv - mx) : 0.f);
      if (lane < V) cand[beam * V + lane] = xv - (mx + logf(sum));
    }
    __syncthreads();
  }
  for (int c = lane; c < n; c += 64) {
    int beam = c / V, v = c - beam * V;
    int row = b * kin + beam;
    float lp = logits ? cand[c] : logp[(int64_t)row * V + v];
    if (prev_tok) {
      int pt = prev_tok[row];
      if (v == 0 && (pt == 1 || pt == 3)) lp = 0.f;             // model.lua:448-449: finished beams continue with PAD at zero cost
      lp += beam_scores[b * kin + beam];                        // model.lua:450
    }
    if (tv.mask && !trie_ok(tv, first ? 0 : tv.loc_in[row], v, first)) lp = -INFINITY;   // model.lua:413,469
    cand[c] = lp;
  }
  __syncthreads();
  float first_best = -INFINITY; int first_bi = 0;
  for (int k = 0; k < kout; ++k) {
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < n; c += 64) { float v = cand[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    // fewer admissible candidates than beams: repeat the best one, model.lua:419-433 (and the intent of :477-497, S11).  The same
    // fallback guards a row of NaN / -inf scores without a dictionary (a diverged model): the index must stay inside the row.
    if (bi == 0x7fffffff) { best = first_best; bi = first_bi; }
    if (k == 0) { first_best = best; first_bi = bi; }
    if (lane == 0) {
      cand[bi] = -INFINITY;
      tokens[b * kout + k] = bi % V + 1;                        // model.lua:456-458
      parents[b * kout + k] = bi / V;                           // model.lua:516 (0-based; S9 fixed at t=1: kin=1 -> 0)
      if (tv.mask) tv.loc_out[b * kout + k] = trie_next(tv, first ? 0 : tv.loc_in[b * kin + bi / V], bi % V, first);   // :434-439,499-507
    }
    __syncthreads();
    if (lane == 0) ((volatile float*)cand)[n + k] = best;       // stash; written back after the loop (beam_scores is also an input)
    __syncthreads();
  }
  for (int k = lane; k < kout; k += 64) beam_scores[b * kout + k] = cand[n + k];
}
// Decode step tail in ONE launch: projector (output_projector.lua:3-8: logits = W_o h + b), LogSoftMax, finished-beam masking,
// score accumulation and top-k selection (model.lua:399-404,446-458,516).  One workgroup per batch element; the kin x V logits are
// 512-long fp32 dot products spread over the 4 waves (lanes stride the hidden dimension), then the selection of beam_select_kernel.
constexpr int PS_NW = 16;                                      // waves per batch element: one logit per wave at a time
__global__ __launch_bounds__(64 * PS_NW) void project_select_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ wo,
                                                             const float* __restrict__ bo, int Hd, const int32_t* __restrict__ prev_tok,
                                                             float* __restrict__ beam_scores, int32_t* __restrict__ tokens,
                                                             int32_t* __restrict__ parents, int kin, int kout, int V, TrieView tv) {
  extern __shared__ float cand[];                               // kin*V (+ kout stash)
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool first = prev_tok == nullptr;
  const int n = kin * V;
#pragma unroll 4
  for (int c = wave; c < n; c += PS_NW) {                       // logits (unrolled: the loads of the next outputs overlap the reductions)
    const int beam = c / V, v = c - beam * V;
    const float* hr = h + (int64_t)(b * kin + beam) * ldh;
    const float* wr = wo + (int64_t)v * Hd;
    float s = 0.f;
    for (int j = lane * 4; j < Hd; j += 256) {
      const float4 a = *reinterpret_cast<const float4*>(hr + j), w4 = *reinterpret_cast<const float4*>(wr + j);
      s = fmaf(a.x, w4.x, s); s = fmaf(a.y, w4.y, s); s = fmaf(a.z, w4.z, s); s = fmaf(a.w, w4.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) cand[c] = s + bo[v];
  }
  __syncthreads();
  if (wave == 0) {                                              // LogSoftMax per beam row (V <= 64), masking, running score
    for (int beam = 0; beam < kin; ++beam) {
      const int row = b * kin + beam;
      const float xv = lane < V ? cand[beam * V + lane] : -INFINITY;
      const float mx = wave_max(xv);
      const float sum = wave_sum(lane < V ? expf(xv - mx) : 0.f);
      float lp = xv - (mx + logf(sum));
      if (prev_tok) {
        const int pt = prev_tok[row];
        if (lane == 0 && (pt == 1 || pt == 3)) lp = 0.f;        // model.lua:448-449
        lp += beam_scores[row];                                 // model.lua:450
      }
      if (tv.mask && lane < V && !trie_ok(tv, first ? 0 : tv.loc_in[row], lane, first)) lp = -INFINITY;   // model.lua:413,469
      if (lane < V) cand[beam * V + lane] = lp;
    }
    float first_best = -INFINITY; int first_bi = 0;
    for (int k = 0; k < kout; ++k) {                            // top-k: descending score, ties -> lowest index (single wave: no barriers)
      float best = -INFINITY; int bi = 0x7fffffff;
      for (int c = lane; c < n; c += 64) { float v = cand[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      // model.lua:419-433: repeat the best admissible candidate; also the guard for NaN / -inf rows (see beam_select_kernel)
      if (bi == 0x7fffffff) { best = first_best; bi = first_bi; }
      if (k == 0) { first_best = best; first_bi = bi; }
      if (lane == 0) {
        cand[bi] = -INFINITY; cand[n + k] = best;
        tokens[b * kout + k] = bi % V + 1;
        parents[b * kout + k] = bi / V;
        if (tv.mask) tv.loc_out[b * kout + k] = trie_next(tv, first ? 0 : tv.loc_in[b * kin + bi / V], bi % V, first);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int k = lane; k < kout; k += 64) beam_scores[b * kout + k] = cand[n + k];
  }
}
void project_select(hipStream_t s, const float* h, int64_t ldh, const float* wo, const float* bo, int Hd, const int32_t* prev_tok,
                    float* beam_scores, int32_t* tokens, int32_t* parents, int B, int kin, int kout, int V, const TrieView* tv) {
  size_t sh = (size_t)(kin * V + kout) * sizeof(float);
  hipLaunchKernelGGL(project_select_kernel, dim3(B), dim3(64 * PS_NW), sh, s, h, ldh, wo, bo, Hd, prev_tok, beam_scores, tokens, parents, kin, kout, V,
                     tv ? *tv : TrieView{});
}
void beam_select(hipStream_t s, const float* logp, const int32_t* prev_tok, float* beam_scores, int32_t* tokens, int32_t* parents,
                 int B, int kin, int kout, int V, const float* logits, int64_t ldl, const TrieView* tv) {
  size_t sh = (size_t)(kin * V + kout) * sizeof(float);
  if (V > 64) logits = nullptr;                                 // the fused LogSoftMax needs one lane per class
  hipLaunchKernelGGL(beam_select_kernel, dim3(B), dim3(64), sh, s, logp, prev_tok, beam_scores, tokens, parents, kin, kout, V, logits, ldl,
                     tv ? *tv : TrieView{});
}
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                                                          int64_t ldd, const int32_t* __restrict__ parents, int B, int kin, int kout,
                                                          int width) {
  int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)B * kout * width) return;
  int c = (int)(id % width); int64_t r = id / width; int b = (int)(r / kout);
  int sr = (kin == 1) ? b : b * kin + parents[r];
  dst[r * ldd + c] = src[(int64_t)sr * lds + c];
}
// dst[r][:] = table[tok[r*stride] - 1][:]  (decode: the embedding part of the first layer's gate input, one table row per token)
__global__ __launch_bounds__(256) void token_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ tok, int64_t stride,
                                                         float* __restrict__ dst, int R, int width4) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)R * width4) return;
  const int c = (int)(id % width4); const int64_t r = id / width4;
  const int v = tok[r * stride] - 1;
  reinterpret_cast<float4*>(dst)[r * width4 + c] = reinterpret_cast<const float4*>(table)[(int64_t)v * width4 + c];
}
void token_rows(hipStream_t s, const float* table, const int32_t* tok, int64_t stride, float* dst, int R, int width) {
  const int64_t n = (int64_t)R * (width / 4);
  hipLaunchKernelGGL(token_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, table, tok, stride, dst, R, width / 4);
}
void gather_beam_rows(hipStream_t s, const float* src, int64_t lds, float* dst, int64_t ldd, const int32_t* parents, int B, int kin, int kout, int width);
// the same gather for up to 8 state tensors of one decode step in ONE launch (model.lua:521-535 gathers c and h of every layer and the
// input feed by the same parents: five dependent ~6 us dispatches per beam step before); 16-byte accesses, blockIdx.y = tensor
struct GatherMany { const float* src[8]; float* dst[8]; bf16_t* dstb[8]; };      // dstb (optional, round 6): a bf16 copy of the gathered rows (the decode chain's shadows)
__global__ __launch_bounds__(256) void gather_rows_many_kernel(GatherMany g, int64_t ld, const int32_t* __restrict__ parents, int B, int kin, int kout, int width4) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= (int64_t)B * kout * width4) return;
  const int c = (int)(id % width4); const int64_t r = id / width4; const int b = (int)(r / kout);
  const int sr = (kin == 1) ? b : b * kin + parents[r];
  const float4 v = reinterpret_cast<const float4*>(g.src[blockIdx.y] + (int64_t)sr * ld)[c];
  reinterpret_cast<float4*>(g.dst[blockIdx.y] + r * ld)[c] = v;
  if (bf16_t* const db = g.dstb[blockIdx.y]) {
    typedef bf16_t bf16x4g __attribute__((ext_vector_type(4)));
    bf16x4g o; o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    reinterpret_cast<bf16x4g*>(db + r * ld)[c] = o;
  }
}
void gather_beam_rows_many(hipStream_t s, int n, const float* const* src, float* const* dst, int64_t ld, const int32_t* parents, int B, int kin, int kout, int width,
                           bf16_t* const* dstb) {
  if (n <= 0) return;
  if (width % 4 || ld % 4 || n > 8) {
    for (int i = 0; i < n; ++i) { gather_beam_rows(s, src[i], ld, dst[i], ld, parents, B, kin, kout, width); if (dstb && dstb[i]) copy2d_bf16(s, dst[i], ld, dstb[i], ld, B * kout, width); }
    return;
  }
  GatherMany g; for (int i = 0; i < 8; ++i) { g.src[i] = src[i < n ? i : 0]; g.dst[i] = dst[i < n ? i : 0]; g.dstb[i] = (dstb && i < n) ? dstb[i] : nullptr; }
  const int64_t items = (int64_t)B * kout * (width / 4);
  hipLaunchKernelGGL(gather_rows_many_kernel, dim3((unsigned)cdiv(items, 256), n), dim3(256), 0, s, g, ld, parents, B, kin, kout, width / 4);
}
void gather_beam_rows(hipStream_t s, const float* src, int64_t lds, float* dst, int64_t ldd, const int32_t* parents, int B, int kin,
                      int kout, int width) {
  int64_t n = (int64_t)B * kout * width;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, src, lds, dst, ldd, parents, B, kin, kout, width);
}
__global__ void backtrace_kernel(const int32_t* __restrict__ hist_tok, const int32_t* __restrict__ hist_par,
                                 const float* __restrict__ beam_scores, int32_t* labels, float* scores, int Lt, int B, int k) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float best = beam_scores[b * k]; int bi = 0;
  for (int i = 1; i < k; ++i) if (beam_scores[b * k + i] > best) { best = beam_scores[b * k + i]; bi = i; }   // torch.max: first max
  scores[b] = best;
  int idx = bi;
  for (int t = Lt - 1; t >= 0; --t) {
    labels[(int64_t)b * Lt + t] = hist_tok[((int64_t)t * B + b) * k + idx];
    idx = hist_par[((int64_t)t * B + b) * k + idx];
  }
}
void beam_backtrace(hipStream_t s, const int32_t* hist_tok, const int32_t* hist_par, const float* beam_scores, int32_t* labels,
                    float* scores, int Lt, int B, int k) {
  hipLaunchKernelGGL(backtrace_kernel, dim3(cdiv(B, 128)), dim3(128), 0, s, hist_tok, hist_par, beam_scores, labels, scores, Lt, B, k);
}
__global__ void fill_i32_kernel(int32_t* p, int32_t v, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
void fill_i32(hipStream_t s, int32_t* p, int32_t v, int64_t n) {
  hipLaunchKernelGGL(fill_i32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, p, v, n);
}

// Levenshtein distance of two id rows, each cut at its first EOS (3): string.levenshtein (utils.lua:55-94) applied to the strings
// evalWordErrRate builds (utils.lua:141-168; numlist2str is injective on ids, so ids compare like the bytes).  One lane per row;
// the single DP row lives in LDS as [j][lane] (conflict-free), the left/diagonal neighbours in registers.
__global__ __launch_bounds__(64) void edit_distance_kernel(const int32_t* __restrict__ labels, const int32_t* __restrict__ targets,
                                                           int B, int L, int32_t* __restrict__ dist, int32_t* __restrict__ target_len) {
  extern __shared__ int32_t dp[];                               // (L+1) * 64
  const int lane = threadIdx.x, b = blockIdx.x * 64 + lane;
  if (b >= B) return;
  const int32_t* p = labels + (int64_t)b * L; const int32_t* g = targets + (int64_t)b * L;
  int lp = 0, lg = 0;
  while (lp < L && p[lp] != 3) ++lp;
  while (lg < L && g[lg] != 3) ++lg;
  for (int j = 0; j <= lg; ++j) dp[j * 64 + lane] = j;          // row i = 0
  for (int i = 1; i <= lp; ++i) {
    const int32_t pc = p[i - 1];
    int diag = dp[lane]; int left = i; dp[lane] = i;
    for (int j = 1; j <= lg; ++j) {
      const int up = dp[j * 64 + lane];
      const int v = min(min(up + 1, left + 1), diag + (pc == g[j - 1] ? 0 : 1));
      dp[j * 64 + lane] = v; diag = up; left = v;
    }
  }
  dist[b] = dp[lg * 64 + lane];
  if (target_len) target_len[b] = lg;
}
void edit_distance(hipStream_t s, const int32_t* labels, const int32_t* targets, int B, int L, int32_t* dist, int32_t* target_len) {
  const size_t lds = (size_t)(L + 1) * 64 * sizeof(int32_t);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)edit_distance_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // L > 255: above the default dynamic-LDS limit
  hipLaunchKernelGGL(edit_distance_kernel, dim3(cdiv(B, 64)), dim3(64), (size_t)(L + 1) * 64 * sizeof(int32_t), s, labels, targets, B, L,
                     dist, target_len);
}

}  // namespace aocr
