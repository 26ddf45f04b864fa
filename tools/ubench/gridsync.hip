// Cost of a grid-wide barrier on MI355X (cooperative launch, one workgroup per CU): the budget of a persistent decoder
// kernel that would replace ~11 dependent launches per decoder step by ~5 barriers.
//  variant 0: cooperative_groups grid.sync()
//  variant 1: hand-rolled sense-free counter barrier: one agent-scope atomic add per workgroup, relaxed polling by lane 0,
//             agent-scope release before / acquire after (monotonic ticket: no reset)
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_cg(int iters, float* buf) {
  cg::grid_group g = cg::this_grid();
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    buf[blockIdx.x * 256 + threadIdx.x] = v;             // something to make visible
    g.sync();
    v += buf[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
  }
  if (v == -1.f) buf[0] = v;
}

__global__ __launch_bounds__(256) void k_ctr(int iters, float* buf, unsigned* ctr) {
  float v = threadIdx.x;
  const unsigned nb = gridDim.x;
  for (int i = 0; i < iters; ++i) {
    buf[blockIdx.x * 256 + threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(i + 1) * nb;
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { if (++spins > (1L << 26)) break; }   // bounded: never hang the box
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    v += buf[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
  }
  if (v == -1.f) buf[0] = v;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int dev = 0, cus = 0; CK(hipGetDevice(&dev)); CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float* buf; unsigned* ctr; CK(hipMalloc(&buf, 1024 * 256 * 4)); CK(hipMalloc(&ctr, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {64, 128, cus}) {
    int iters = 2000;
    void* args[] = {&iters, &buf};
    CK(hipLaunchCooperativeKernel((void*)k_cg, dim3(blocks), dim3(256), args, 0, 0)); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((void*)k_cg, dim3(blocks), dim3(256), args, 0, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("grid.sync()         %3d workgroups: %6.2f us per barrier\n", blocks, ms * 1e3 / iters);
    CK(hipMemset(ctr, 0, 4));
    void* args2[] = {&iters, &buf, &ctr};
    CK(hipLaunchCooperativeKernel((void*)k_ctr, dim3(blocks), dim3(256), args2, 0, 0)); CK(hipDeviceSynchronize());
    CK(hipMemset(ctr, 0, 4));
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((void*)k_ctr, dim3(blocks), dim3(256), args2, 0, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("counter barrier     %3d workgroups: %6.2f us per barrier\n", blocks, ms * 1e3 / iters);
  }
  return 0;
}
