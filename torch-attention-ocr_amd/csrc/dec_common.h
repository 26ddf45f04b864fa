// dec_common.h -- device helpers shared by the whole-sequence decoder kernels (dec_cluster.hip: one chain of 32 rows per group;
// dec_chain.hip: two interleaved chains of 16 rows): L1-bypassing loads, exchange stores, counted vmcnt waits, DPP wave reductions.
#pragma once
#include "ops.h"

namespace aocr {

// two-chain, tag-free-exchange form of the decoder kernels (dec_chain.hip); AOCR_NO_DEC_CHAINS=1 keeps dec_cluster.hip's kernels
bool dec_chain_enabled();
void dec_chain_forward(hipStream_t s, const DecClFwdArgs& a, bool greedy_decode = false);
void dec_chain_backward(hipStream_t s, const DecClBwdArgs& a);

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

namespace {
constexpr int DC_SPIN_LIMIT = 1 << 18;
constexpr int HD = 512, NM = 32, R = 32, PA = HD * 2 + 16;      // hidden size, members per group, rows per group, LDS operand pitch
constexpr int PZ = 4096 + 16;                                      // LDS pitch of a d z operand row (2048 bf16)
constexpr int LDS_BYTES = 3 * R * PA + 32768 + 4736 + 10240;     // three operand buffers + the reduction scratch + the decode scratch (+ its gate-input table slice)

// ---- VMEM in program order: polls first, the previous phase's output stores behind them, then `s_waitcnt vmcnt(#stores)` -- the
// polls are waited for, the stores are not (vmcnt counts loads and stores in issue order on gfx9).  Every store below is ONE
// instruction whatever the lane's predicate (invalid lanes point at a trash slot), so the counts are exact.
// L1-bypassing loads (sc1): scalar base + 32-bit lane offset
// (group inside one XCD: its L2 is the point of coherence; otherwise system scope on both sides)
__device__ __forceinline__ void ld16_sc1(u32x4& v, unsigned voff, const void* sbase, bool local) {
  if (local) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void ld4_sc1(unsigned& v, unsigned voff, const void* sbase, bool local) {
  if (local) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  else asm volatile("global_load_dword %0, %1, %2 sc0 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
}
// exchange payload / flag stores
__device__ __forceinline__ void pst4(void* p, unsigned v, bool local) {
  if (local) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void pst8(void* p, u32x2 v, bool local) {
  if (local) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 0" ::"v"(p), "v"(v) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dpin(u32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void st4(void* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st4f(void* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st8(void* p, u32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st16f(void* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
// workgroup barrier that orders LDS only: global loads / stores stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
__device__ __forceinline__ unsigned bfbits(float x) { bf16_t h = (bf16_t)x; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
__device__ __forceinline__ unsigned bfpair(float a, float b) { return bfbits(a) | (bfbits(b) << 16); }
__device__ __forceinline__ u64 ldg64(const u64* p) { return __hip_atomic_load(const_cast<u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stg64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-wide reductions on the DPP network (no LDS round trips): butterflies inside a row of 16, then row_bcast15 / row_bcast31
template <int CTRL, int ROWS> __device__ __forceinline__ float dppf(float ident, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
template <class OP> __device__ __forceinline__ float wave_reduce(float v, float ident, OP op) {
  v = op(v, dppf<0xB1, 0xF>(ident, v));          // quad_perm [1,0,3,2]
  v = op(v, dppf<0x4E, 0xF>(ident, v));          // quad_perm [2,3,0,1]
  v = op(v, dppf<0x141, 0xF>(ident, v));         // row_half_mirror
  v = op(v, dppf<0x140, 0xF>(ident, v));         // row_mirror: every lane of a row holds the row's result
  v = op(v, dppf<0x142, 0xA>(ident, v));         // row_bcast15 into rows 1, 3
  v = op(v, dppf<0x143, 0xC>(ident, v));         // row_bcast31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
}  // namespace

}  // namespace aocr
