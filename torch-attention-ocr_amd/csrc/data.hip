// data.hip -- device side of the reference's data path (src/data/data_gen.lua:68-79):
// 255 * rgb2y of a decoded image, then image.scale(img, imgW, 32) -- rows to the target width first, then columns to the
// target height, enlarging by linear interpolation with scale (src-1)/(dst-1), shrinking by averaging the covered source span
// (torch/image generic/image.c: rgb2y, scaleBilinear -> scaleLinear_rowcol; restated in oracle/data_oracle.py).
// One thread per output pixel; every float operation is an explicitly rounded (non-contracted) single-precision op in the
// order of the restatement, so the result is bit-identical to it.  HBM-bound and tiny: the images of a batch are a few MB.
#include "ops.h"

// HIP's __fmul_rn / __fadd_rn are plain operators and hipcc contracts a*b+c into an FMA by default (-ffp-contract=fast, which
// also ignores `#pragma clang fp contract`): this file is compiled with -ffp-contract=off (Makefile: FLAGS_data) so that every
// operation below rounds exactly like the single-precision restatement.

namespace aocr {

namespace {

struct Gray {                                            // 255 * rgb2y of source pixel (row, col), interleaved uint8 HWC, C in {1, 3}
  const uint8_t* img; int w, c;
  __device__ __forceinline__ float at(int row, int col) const {
    const uint8_t* p = img + ((int64_t)row * w + col) * c;
    if (c == 1) return (float)p[0];
    const float k = 1.0f / 255.0f;
    const float r = __fmul_rn((float)p[0], k), g = __fmul_rn((float)p[1], k), b = __fmul_rn((float)p[2], k);
    const float y = __fadd_rn(__fadd_rn(__fmul_rn(0.299f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));
    return __fmul_rn(255.0f, y);
  }
};

// element di of scaleLinear_rowcol(src[0..n), dst_len); F(i) yields src[i]
template <class F> __device__ __forceinline__ float scale_elem(const F& src, int n, int dst_len, int di) {
  if (dst_len == n) return src(di);
  if (dst_len > n) {
    if (n == 1) return src(0);
    if (di == dst_len - 1) return src(n - 1);
    const float scale = __fdiv_rn((float)(n - 1), (float)(dst_len - 1));
    const float sf = __fmul_rn((float)di, scale);
    const int si = (int)sf;
    const float fr = __fsub_rn(sf, (float)si);
    return __fadd_rn(__fmul_rn(__fsub_rn(1.0f, fr), src(si)), __fmul_rn(fr, src(si + 1)));
  }
  const float scale = __fdiv_rn((float)n, (float)dst_len);
  const float s0 = __fmul_rn((float)di, scale), s1 = __fmul_rn((float)(di + 1), scale);
  const int si0_i = (int)s0, si1_i = (int)s1;
  const float si0_f = __fsub_rn(s0, (float)si0_i), si1_f = __fsub_rn(s1, (float)si1_i);
  float acc = __fmul_rn(__fsub_rn(1.0f, si0_f), src(si0_i));
  float cnt = __fsub_rn(1.0f, si0_f);
  for (int si = si0_i + 1; si < si1_i; ++si) { acc = __fadd_rn(acc, src(si)); cnt = __fadd_rn(cnt, 1.0f); }
  if (si1_i < n) { acc = __fadd_rn(acc, __fmul_rn(si1_f, src(si1_i))); cnt = __fadd_rn(cnt, si1_f); }
  return __fdiv_rn(acc, cnt);
}

__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ src, const aocr_image_desc* __restrict__ desc,
                                                         int out_h, int out_w, float* __restrict__ out) {
  const int img = blockIdx.y;
  const int id = blockIdx.x * 256 + threadIdx.x;
  if (id >= out_h * out_w) return;
  const int y = id / out_w, x = id - y * out_w;
  const aocr_image_desc d = desc[img];
  Gray g; g.img = src + d.offset; g.w = d.width; g.c = d.channels;
  // column pass over the row-scaled intermediate tmp[row][x] = scale_elem(gray row, width -> out_w)[x]
  auto tmp = [&](int row) { return scale_elem([&](int col) { return g.at(row, col); }, d.width, out_w, x); };
  out[((int64_t)img * out_h + y) * out_w + x] = scale_elem(tmp, d.height, out_h, y);
}

}  // namespace

void preprocess_lines(hipStream_t s, const uint8_t* src, const aocr_image_desc* desc, int n_images, int out_h, int out_w, float* out) {
  if (n_images <= 0) return;
  hipLaunchKernelGGL(preprocess_kernel, dim3(cdiv(out_h * out_w, 256), n_images), dim3(256), 0, s, src, desc, out_h, out_w, out);
}

}  // namespace aocr
