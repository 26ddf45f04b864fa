// Test infrastructure (tests/test_cotenant_gpu.py): a persistent CO-TENANT kernel with the footprint of a collective -- N workgroups of
// 512 threads that stay resident for a given time, streaming a buffer (copy loop) -- launched on the library's exchange stream from the
// all-reduce callback, so that it runs ACROSS the whole-sequence ("cluster") kernels of the backward pass the way an RCCL all-reduce of
// gradient bucket 0 would.  A collective's kernel does not leave its compute units until its peers have joined; this one does not leave
// until its time is up.  Not part of libaocr.so.
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(512) void cotenant_kernel(float4* __restrict__ buf, size_t n4, unsigned long long ticks, unsigned long long* __restrict__ stamps) {
  const unsigned long long t0 = wall_clock64();                 // 100 MHz
  if (threadIdx.x == 0) stamps[2 * blockIdx.x] = t0;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  while (wall_clock64() - t0 < ticks) {
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {                              // copy loop: read, touch, write back (as a reduce-scatter step does)
      float4 v = buf[i % n4];
      acc.x += v.x; v.y += 1.f;
      buf[i % n4] = v;
      i += stride;
    }
  }
  if (acc.x == 123.456f) buf[0] = acc;                          // keep the loads
  if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}

extern "C" int cotenant_launch(void* stream, void* buf, size_t bytes, int workgroups, double microseconds, void* stamps) {
  hipLaunchKernelGGL(cotenant_kernel, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, (float4*)buf, bytes / 16,
                     (unsigned long long)(microseconds * 100.0), (unsigned long long*)stamps);
  return (int)hipGetLastError();
}
