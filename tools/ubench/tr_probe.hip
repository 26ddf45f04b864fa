// Probes the lane/element mapping of ds_read_b64_tr_b16 (gfx950): LDS image [64 rows][64 cols] of shorts, value = row*100 + col.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
  __syncthreads();
  int lane = threadIdx.x;
  int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  short* addr = &lds[(4 * g + q) * 64 + 4 * p];         // group g: rows 4g..4g+3, cols 0..15
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short h[64 * 64], o[256]; for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = r * 100 + c;
  short *d, *e; hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  return 0;
}
