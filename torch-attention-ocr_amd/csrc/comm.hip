// comm.hip -- the ONE exchange step of the data-parallel path inside the library: the sum over ranks of the flat
// gradient vector between feval and the per-group clip (optim_sgd.lua:38 -> :40), plus the per-channel BatchNorm sums when
// synchronised BatchNorm is on (cnn.lua:23,32,41: the reference's batch statistics are those of the WHOLE batch).
//
// Two providers behind one call: RCCL (ncclAllReduce over xGMI; librccl is bound with dlopen, so a single-GPU host never needs it)
// or a host callback (the Python mirror passes torch.distributed through it: gloo in the tests, nccl = RCCL otherwise).
#include "model.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace aocr {

namespace {
struct NcclId { char internal[128]; };
typedef int (*GetUniqueIdFn)(NcclId*);
typedef int (*CommInitRankFn)(void**, int, NcclId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef int (*CommSplitFn)(void*, int, int, void**, void*);
typedef const char* (*GetErrorStringFn)(int);
struct Rccl { void* h = nullptr; GetUniqueIdFn uid; CommInitRankFn init; AllReduceFn allreduce; CommDestroyFn destroy; GetErrorStringFn errstr; CommSplitFn split; };
Rccl g_rccl;
const char* load_rccl() {
  if (g_rccl.h) return nullptr;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }        // the copy the process already has, if any
  for (int i = 0; i < 3 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return "librccl.so not found";
  g_rccl.uid = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId"); g_rccl.init = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
  g_rccl.allreduce = (AllReduceFn)dlsym(h, "ncclAllReduce"); g_rccl.destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
  g_rccl.errstr = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
  g_rccl.split = (CommSplitFn)dlsym(h, "ncclCommSplit");                       // optional (NCCL >= 2.18 API): second communicator for the BatchNorm sums
  if (!g_rccl.uid || !g_rccl.init || !g_rccl.allreduce || !g_rccl.destroy) return "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy";
  g_rccl.h = h;
  return nullptr;
}
constexpr int NCCL_SUM = 0, NCCL_F32 = 7, NCCL_F64 = 8;

// A whole-sequence kernel that gave up waiting for its group leaves a code in cl_err[0]; the optimizers skip the update on it (device-side
// predicate) and the host repeats the step.  Under data parallelism that decision must be the SAME on every rank: the peers would otherwise
// apply an update computed from a sum that contains this rank's invalid gradients while this rank keeps its parameters (the replicas
// diverge), and a rank-local repeat would issue collectives no peer matches.  So the flag travels with the exchange: 1.0 per rank whose
// code is non-zero, summed, and a rank that sees a positive sum with a clean local code takes AOCR_CL_PEER_TIMEOUT.
constexpr int AOCR_CL_PEER_TIMEOUT = 0x7e;
__global__ void cl_flag_pack_kernel(const int* __restrict__ err, float* __restrict__ flag) { flag[0] = err[0] != 0 ? 1.f : 0.f; }
__global__ void cl_flag_merge_kernel(int* __restrict__ err, const float* __restrict__ flag) { if (flag[0] > 0.f && err[0] == 0) err[0] = AOCR_CL_PEER_TIMEOUT; }
}  // namespace

const char* comm_unique_id(char id[128]) {
  if (const char* e = load_rccl()) return e;
  NcclId u; if (g_rccl.uid(&u) != 0) return "ncclGetUniqueId failed";
  memcpy(id, u.internal, 128);
  return nullptr;
}

static const char* comm_common_init(aocr_model* m, int nranks, int sync_bn) {
  m->comm.nranks = nranks; m->comm.sync_bn = sync_bn != 0;
  if (!m->comm.stream && hipStreamCreateWithFlags(&m->comm.stream, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed";
  if (!m->comm.done && hipEventCreateWithFlags(&m->comm.done, hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
  if (!m->comm.wait0 && (hipEventCreate(&m->comm.wait0) != hipSuccess || hipEventCreate(&m->comm.wait1) != hipSuccess)) return "hipEventCreate failed";
  return nullptr;
}
const char* comm_init_rccl(aocr_model* m, const char id[128], int nranks, int rank, int sync_bn) {
  if (const char* e = load_rccl()) return e;
  NcclId u; memcpy(u.internal, id, 128);
  void* c = nullptr;
  const int rc = g_rccl.init(&c, nranks, u, rank);
  if (rc != 0) { static thread_local char buf[160]; snprintf(buf, sizeof buf, "ncclCommInitRank: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "error"); return buf; }
  m->comm.rccl = c; m->comm.provider = 1;
  // A communicator executes its operations in the order the host issued them.  The BatchNorm sums are issued from inside the forward /
  // backward pass, the gradient buckets after the whole backward pass has been enqueued: on ONE communicator bucket 0 (ready 45 % into
  // the backward pass) would queue behind the last BatchNorm-backward sum (near its end) and the overlap would be lost.  So the
  // BatchNorm sums get a communicator of their own (same ranks, ncclCommSplit); without that entry point they share the first one.
  m->comm.rccl_bn = nullptr;
  // ncclCommSplit is collective: a failure on a SUBSET of the ranks would leave some summing BatchNorm statistics on the second
  // communicator and the others on the first -- mismatched collectives, a hang at the first SyncBN sum.  So a failed split is an error on
  // the rank that sees it (the host aborts the job), never a silent fall-back; only a librccl WITHOUT the entry point (the same on every
  // rank of a node: one library) or AOCR_ONE_COMM=1 (set for the whole job) share one communicator.
  if (sync_bn && g_rccl.split && !getenv("AOCR_ONE_COMM")) {
    void* c2 = nullptr;
    const int rs = g_rccl.split(c, 0, rank, &c2, nullptr);
    if (rs != 0 || !c2) {
      static thread_local char buf[200]; snprintf(buf, sizeof buf, "ncclCommSplit (second communicator for the BatchNorm sums) failed: %s; set AOCR_ONE_COMM=1 on EVERY rank to share one communicator", rs != 0 && g_rccl.errstr ? g_rccl.errstr(rs) : "no communicator returned");
      g_rccl.destroy(c); m->comm.rccl = nullptr; m->comm.provider = 0;
      return buf;
    }
    m->comm.rccl_bn = c2;
  }
  return comm_common_init(m, nranks, sync_bn);
}
const char* comm_init_callback(aocr_model* m, aocr_allreduce_fn fn, void* user, int nranks, int sync_bn) {
  m->comm.fn = fn; m->comm.user = user; m->comm.provider = 2;
  return comm_common_init(m, nranks, sync_bn);
}
void comm_destroy(aocr_model* m) {
  if (m->comm.provider == 1 && m->comm.rccl_bn && g_rccl.h) g_rccl.destroy(m->comm.rccl_bn);
  if (m->comm.provider == 1 && m->comm.rccl && g_rccl.h) g_rccl.destroy(m->comm.rccl);
  if (m->comm.stream) hipStreamDestroy(m->comm.stream);
  if (m->comm.done) hipEventDestroy(m->comm.done);
  if (m->comm.wait0) hipEventDestroy(m->comm.wait0);
  if (m->comm.wait1) hipEventDestroy(m->comm.wait1);
  m->comm = CommState{};
}

// in-place sum over ranks of `count` elements (dtype 0 = fp32, 1 = fp64) enqueued on `stream`; channel 0 = gradient exchange,
// 1 = BatchNorm sums (a communicator / process group of its own: see comm_init_rccl)
int comm_allreduce(aocr_model* m, void* buf, int64_t count, int dtype, hipStream_t stream, int channel) {
  if (m->comm.nranks <= 1 && m->comm.provider != 1) return 0;
  if (m->comm.provider == 1)
    return g_rccl.allreduce(buf, buf, (size_t)count, dtype ? NCCL_F64 : NCCL_F32, NCCL_SUM, (channel && m->comm.rccl_bn) ? m->comm.rccl_bn : m->comm.rccl, stream);
  if (m->comm.provider == 2) return m->comm.fn(m->comm.user, buf, count, dtype | (channel ? AOCR_COMM_CHANNEL_BN : 0), (void*)stream);
  return 0;
}

// Bucketed sum of the gradient vector on the library's second stream: every bucket waits for the event the backward pass recorded
// when that part of the vector was complete (backward_all), so the exchange of the decoder / encoder / upper-CNN buckets runs beside
// the rest of the backward pass; the model's stream joins at the end (the clip needs every bucket).
int comm_allreduce_grads(aocr_model* m, float* loss_dev) {
  if (m->comm.provider == 0) return 0;
  hipStream_t cs = m->comm.stream;
  int64_t b[AOCR_GRAD_BUCKETS], e[AOCR_GRAD_BUCKETS];
  aocr_grad_buckets(&m->cfg, b, e);
  for (int k = 0; k < AOCR_GRAD_BUCKETS; ++k) {
    if (hipStreamWaitEvent(cs, m->grad_ev[k], 0) != hipSuccess) return 1;
    if (comm_allreduce(m, m->grads + b[k], e[k] - b[k], 0, cs, 0) != 0) return 2;
    if (k == 0 && loss_dev && comm_allreduce(m, loss_dev, 1, 0, cs, 0) != 0) return 2;      // the loss is final before the backward pass starts
    if (k == 1 && m->cl_err) {
      // every whole-sequence kernel of the step has completed (grad_ev[1] is recorded behind the encoder BPTT): agree on the time-out flag
      float* flag = reinterpret_cast<float*>(m->cl_err + 8);                               // ints 8..11 of the flag block: exchange scratch (1..4 hold the encoder kernel's time-out diagnostics, the trash slots start at 16)
      hipLaunchKernelGGL(cl_flag_pack_kernel, dim3(1), dim3(1), 0, cs, m->cl_err, flag);
      if (comm_allreduce(m, flag, 1, 0, cs, 0) != 0) return 2;
      hipLaunchKernelGGL(cl_flag_merge_kernel, dim3(1), dim3(1), 0, cs, m->cl_err, flag);
    }
  }
  // the model's stream joins here.  wait0 fires when the backward pass is done, wait1 when the last bucket is: their distance is the
  // part of the exchange the backward pass did NOT hide (aocr_comm_exposed_ms)
  if (hipEventRecord(m->comm.done, cs) != hipSuccess) return 1;
  hipEventRecord(m->comm.wait0, m->s);
  if (hipStreamWaitEvent(m->s, m->comm.done, 0) != hipSuccess) return 1;
  hipEventRecord(m->comm.wait1, m->s); m->comm.timed = true;
  return 0;
}

}  // namespace aocr
