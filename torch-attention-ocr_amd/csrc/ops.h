// ops.h -- host-side launchers of every kernel on the hot path (all enqueue on a stream, none synchronise).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "aocr.h"
#include "epilogues.h"
#include "mfma_gemm.h"

namespace aocr {

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- helpers to build loaders
inline LoadK make_loadk(const float* p, int64_t ld, int rows, int K) {
  LoadK l; l.p0 = p; l.ld0 = ld; l.K0 = K; l.p1 = nullptr; l.ld1 = 0; l.rows = rows; l.K = K;
  l.vec = (((uintptr_t)p & 15) == 0) && (ld % 4 == 0);
  return l;
}
inline LoadK make_loadk2(const float* p0, int64_t ld0, int K0, const float* p1, int64_t ld1, int K1, int rows) {
  LoadK l; l.p0 = p0; l.ld0 = ld0; l.K0 = K0; l.p1 = p1; l.ld1 = ld1; l.rows = rows; l.K = K0 + K1;
  l.vec = (((uintptr_t)p0 & 15) == 0) && (ld0 % 4 == 0) && (((uintptr_t)p1 & 15) == 0) && (ld1 % 4 == 0) && (K0 % 16 == 0);
  return l;
}
inline LoadMN make_loadmn(const float* p, int64_t ld, int rows, int K) {
  LoadMN l; l.p = p; l.ld = ld; l.rows = rows; l.K = K;
  l.vec = (((uintptr_t)p & 15) == 0) && (ld % 4 == 0);
  return l;
}
inline EpStore make_store(float* C, int64_t ldc, int M, int N, const float* bias = nullptr, const float* bias2 = nullptr,
                          int flags = 0) {
  EpStore e; e.C = C; e.ldc = ldc; e.M = M; e.N = N; e.bias = bias; e.bias2 = bias2; e.flags = flags;
  e.C1 = nullptr; e.ldc1 = 0; e.N0 = N; return e;
}

// ---- MFMA contractions (ops_gemm.hip)
// ksplit > 1 requires ep.flags & EP_ATOMIC.
void launch_big_kk(hipStream_t s, bool bf16, const LoadK& a, const LoadK& b, const EpStore& ep, int M, int N, int K, int ksplit);
void launch_big_kmn(hipStream_t s, bool bf16, const LoadK& a, const LoadMN& b, const EpStore& ep, int M, int N, int K, int ksplit);
void launch_big_mnmn(hipStream_t s, bool bf16, const LoadMN& a, const LoadMN& b, const EpStore& ep, int M, int N, int K, int ksplit);
void launch_conv_fwd(hipStream_t s, bool bf16, const LoadConvK& a, const LoadK& b, const EpConv& ep, int M, int N, int K);
void launch_conv_dgrad(hipStream_t s, bool bf16, const LoadConvK& a, const LoadConvWT& b, const EpStore& ep, int M, int N, int K);
void launch_conv_wgrad(hipStream_t s, bool bf16, const LoadMN& a, const LoadConvXcol& b, const EpStore& ep, int M, int N, int K, int ksplit);

typedef SmallArgs<LoadK, LoadK, EpGatesFwd> GatesFwdArgs;
typedef SmallArgs<LoadK, LoadK, EpStore> SmallKKArgs;
typedef SmallArgs<LoadK, LoadMN, EpStore> SmallKMNArgs;
typedef SmallArgs<LoadK, LoadMN, EpGatesBwd> GatesBwdArgs;
// nz = 1 or 2 argument sets (blockIdx.z); M rows, H (or N) columns.
void launch_small_gates_fwd(hipStream_t s, bool bf16, int nz, const GatesFwdArgs* z, int M, int H);
void launch_small_kk(hipStream_t s, bool bf16, int nz, const SmallKKArgs* z, int M, int N);
void launch_small_kmn(hipStream_t s, bool bf16, int nz, const SmallKMNArgs* z, int M, int N);
void launch_small_gates_bwd(hipStream_t s, bool bf16, int nz, const GatesBwdArgs* z, int M, int H);
typedef SmallArgs<LoadK, LoadK, EpGatesBwd> GatesBwdKKArgs;     // fp32 mode, B = fp32 transposed weights (K-contiguous)
void launch_small_gates_bwd_kk(hipStream_t s, int nz, const GatesBwdKKArgs* z, int M, int H);
// bf16 mode with bf16 weight shadows as the B operand (always K-contiguous: W for y = x W^T, W^T for y = x W)
typedef SmallArgs<LoadK, LoadKh2, EpGatesFwd> GatesFwdArgsH;
typedef SmallArgs<LoadK, LoadKh2, EpStore> SmallArgsH;
typedef SmallArgs<LoadK, LoadKh2, EpGatesBwd> GatesBwdArgsH;
typedef SmallArgs<LoadKh2, LoadKh2, EpGatesFwd> GatesFwdArgsHH;    // A read from its bf16 shadow as well
typedef SmallArgs<LoadKh2, LoadKh2, EpStore> SmallArgsHH;
typedef SmallArgs<LoadKh2, LoadKh2, EpGatesBwd> GatesBwdArgsHH;
void launch_small_gates_fwd_hh(hipStream_t s, int nz, const GatesFwdArgsHH* z, int M, int H);
// recurrent-step products at large batch (>= 320 rows, >= 1024 columns): 128 x 128 LDS-DMA tiles + an elementwise cell pass; false: shape not taken (ops_gemm.hip)
bool big_step_store(hipStream_t s, const LoadKh2& a, const LoadKh2& b, const EpStore& ep, int M, int N);
bool big_step_gates_fwd(hipStream_t s, const LoadKh2& a, const LoadKh2& b, const EpGatesFwd& ep, int M, int H, float* zbuf, size_t zbuf_floats);
void launch_small_hh(hipStream_t s, int nz, const SmallArgsHH* z, int M, int N);
void launch_small_gates_bwd_hh(hipStream_t s, int nz, const GatesBwdArgsHH* z, int M, int H);
void launch_small_gates_fwd_h(hipStream_t s, int nz, const GatesFwdArgsH* z, int M, int H);
void launch_small_h(hipStream_t s, int nz, const SmallArgsH* z, int M, int N);
void launch_small_gates_bwd_h(hipStream_t s, int nz, const GatesBwdArgsH* z, int M, int H);
inline LoadKh2 make_loadkh(const bf16_t* p, int64_t ld, int rows, int K) {
  LoadKh2 l; l.p0 = p; l.ld0 = ld; l.K0 = K; l.p1 = nullptr; l.ld1 = 0; l.rows = rows; l.K = K; return l;
}
inline LoadKh2 make_loadkh2(const bf16_t* p0, int64_t ld0, int K0, const bf16_t* p1, int64_t ld1, int K1, int rows) {
  LoadKh2 l; l.p0 = p0; l.ld0 = ld0; l.K0 = K0; l.p1 = p1; l.ld1 = ld1; l.rows = rows; l.K = K0 + K1; return l;
}
// wb [R][C] = bf16(w[r*ld + c]), wtb [C][R] = its transpose (dense)
void weight_shadows(hipStream_t s, const float* w, int64_t ld, int R, int C, bf16_t* wb, bf16_t* wtb);
void transpose_f32(hipStream_t s, const float* w, int64_t ld, int R, int C, float* wt);                    // wt [C][R]
void conv_weight_transpose_f32(hipStream_t s, const float* w, float* wt, int Cout, int KK, int Cin);     // wt [Cin][KK][Cout]

// generic GEMM used by the C ABI and the hoisted projections; picks ksplit when allowed (atomic accumulate).
// projector + LogSoftMax / ClassNLL + d logits + projector data gradient of all L B rows as one launch (ops_gemm.hip: project_loss_kernel; bf16 mode, V <= 40);
// every output bit-identical to gemm (skinny) + logsoftmax_nll + gemm (K = V)
bool project_loss_ok(int rows, int V, int Hd);
void project_loss(hipStream_t s, const float* out, int64_t ldo, const float* wo, const float* bo, float* logits, float* dlogits, int64_t ld, float* nll, float* dout,
                  const int32_t* tgt, int64_t st, int64_t sb, int Bt, int rows, int V, int Hd, float scale);
int gemm(hipStream_t s, bool bf16, const float* A, int64_t lda, bool a_kmajor, const float* B, int64_t ldb, bool b_kmajor,
          float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* bias2, int flags);

// Grouped weight-gradient contractions C_i[M_i,N_i] += A_i^T B_i (A_i [K_i][M_i], B_i [K_i][N_i], all M/N-contiguous fp32):
// one launch for up to 8 problems; K is split only as far as needed to fill the chip once.
struct WGradProblem { const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc; int M, N, K;
                      const bf16_t* Ab = nullptr; const bf16_t* Bb = nullptr; };     // optional bf16 shadows of A and B (same lda/ldb)
void grouped_wgrad(hipStream_t s, bool bf16, const WGradProblem* p, int n, float* part = nullptr, size_t part_floats = 0);     // part: split-K slab scratch (enables the LDS-DMA kernel for deep problems)
void kprobe_read(unsigned long long out[8]);   // debugging probe of the tagged halo kernels (mfma_gemm.h: g_kprobe)

// C = A B^T (+bias) with both operands read from K-contiguous bf16 shadows (A [M][K], B [N][K])
bool gemm_hh_cat(hipStream_t s, const bf16_t* A0, const bf16_t* A1, int64_t lda, const bf16_t* B0, const bf16_t* B1, int64_t ldb, float* C, int64_t ldc, int M, int N, int K0, int K1);   // C = [A0 | A1] . [B0 | B1]^T, one launch; false: shape not taken
void gemm_hh(hipStream_t s, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
             const float* bias, const float* bias2, int flags);
// the same with a bf16 copy of C written beside it (plain stores)
void gemm_hh_shadow(hipStream_t s, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, bf16_t* Cb, int64_t ldcb,
                    int M, int N, int K);

// ---- convolution layers (ops_gemm.hip)
// xb/wb/dyb/wtb: optional bf16 shadows of the operands (both of a contraction's operands must be given to take the
// bf16-source path); yb: optional bf16 shadow of the output to write.
void conv_forward(hipStream_t s, bool bf16, const float* x, const float* w, const float* bias, float* y, uint8_t* idx,
                  int B, int H, int W, int Cin, int Cout, int ks, int pad, int relu, int pool,
                  const bf16_t* xb = nullptr, const bf16_t* wb = nullptr, bf16_t* yb = nullptr, int profile_tag = 0,
                  const float* bn_save = nullptr, const float* bn_w = nullptr, const float* bn_b = nullptr,
                  double* bn_part = nullptr, int* bn_chunks = nullptr, int* y_bf16 = nullptr);
// bn_part / bn_chunks: scratch of the training-mode BatchNorm that follows (bn_relu_forward's `scratch`); when the launch taken stages its output tiles through
// LDS the per-tile column sums are written there and *bn_chunks = the number of row tiles (pass it to bn_relu_forward as stats_chunks), else *bn_chunks = 0.
// y_bf16 (with them): the caller accepts y written as bf16 INTO THE SAME BUFFER (half of it); *y_bf16 = 1 when that happened (only under AOCR_BN_Y16=1: see conv_forward)
// bn_save != nullptr: evaluation-mode BatchNorm + ReLU folded into the conv epilogue (bn_save from bn_eval_prepare)
void bn_eval_prepare(hipStream_t s, const float* rm, const float* rv, float* save, int C);
// profile_tag != 0: the same kernel under a distinct symbol (aocr_profile_kernel), so profilers list these launches separately
struct BnBwdFuse { const float* x; const bf16_t* yb; const float* save; double* part; };   // pre-BatchNorm map (fp32), post-ReLU bf16 shadow (mask), {mean, invstd}, bn_relu_backward's scratch
void conv_backward_data(hipStream_t s, bool bf16, const float* dy, const float* w, float* dx, int B, int H, int W, int Cin,
                        int Cout, int ks, int pad, const bf16_t* dyb = nullptr, const bf16_t* wtb = nullptr, const float* wtf = nullptr,
                        int* dx16 = nullptr /* the caller accepts dx written as bf16 into the first half of the same buffer; *dx16 = 1 when that happened */,
                        const struct BnBwdFuse* bnb = nullptr, int* bnb_chunks = nullptr /* the BatchNorm backward behind this data gradient: its partial sums from the conv epilogue; *bnb_chunks = row tiles written (pass to bn_relu_backward as sums_chunks), 0 = not taken */);
void conv_backward_filter(hipStream_t s, bool bf16, const float* x, const float* dy, float* dw, float* dbias, int B, int H,
                          int W, int Cin, int Cout, int ks, int pad, const bf16_t* xb = nullptr, const bf16_t* dyb = nullptr,
                          float* part = nullptr, size_t part_floats = 0, int profile_tag = 0);   // part: scratch for the split-K slabs (else fp32 atomics)
void splitk_reduce(hipStream_t s, const float* part, int ks, size_t n, float* out);   // out[i] += sum_z part[z * n + i]  (n % 4 == 0)

// ---- everything that is not a contraction (ops_misc.hip)
// route (conv1_route_elems() uint16): the pooling/ReLU decision of every window, four 4-bit codes per word, for conv1_backward
void conv1_forward(hipStream_t s, const float* x, const float* w, const float* bias, float* y, int B, int H, int W,
                   bf16_t* yb = nullptr, uint16_t* route = nullptr);
size_t conv1_route_elems(int B, int H, int W);
void conv_weight_shadows(hipStream_t s, const float* w, bf16_t* wb, bf16_t* wtb, int Cout, int KK, int Cin);
struct ColsumJobs;
void conv1_backward(hipStream_t s, const float* x, const float* w, const float* bias, const float* dyp, float* dw, float* db,
                    int B, int H, int W, float* scratch = nullptr, ColsumJobs* defer = nullptr,
                    const uint16_t* route = nullptr /* conv1_forward's decisions for the SAME x, w, bias; null: re-evaluated here */);
void unpool_relu_backward(hipStream_t s, const float* dpooled, const float* pooled, const uint8_t* idx, float* dy, int B,
                          int Ho, int Wo, int C, int pool, bf16_t* dyb = nullptr, float* dbias = nullptr,
                          float* partial = nullptr, const bf16_t* pooledb = nullptr, ColsumJobs* defer = nullptr,
                          const bf16_t* dpooled16 = nullptr /* d(pooled) as bf16 instead of fp32 (bf16-mode 8-channel kernel only: conv_backward_data's dx16) */);
// defer: the column sum that finishes a partial slab is queued (colsum_flush) instead of launched -- the slab must then stay untouched until the flush
// dbias + partial (>= 2048*C floats scratch): fused bias gradient, dy may then be null; pooledb: bf16 shadow of pooled (mask source)
void bf16_to_f32(hipStream_t s, const bf16_t* src, float* dst, int64_t n);
size_t bn_scratch_bytes(int C);
// synchronised BatchNorm: sums `count` elements (dtype 1 = fp64) of a device buffer over the data-parallel ranks, on stream s
struct BnSync { int (*allreduce)(void* ctx, void* buf, int64_t count, int dtype, hipStream_t s); void* ctx; };
void bn_relu_forward(hipStream_t s, const float* x, float* y, const float* w, const float* b, float* rm, float* rv,
                     float* save, void* scratch, int64_t rows, int C, int training, int update_running, int tb_rows,
                     bf16_t* yb = nullptr, const BnSync* sync = nullptr, int stats_chunks = 0 /* > 0: scratch already holds that many chunks of partial sums (conv_forward's bn_part) */,
                     const bf16_t* xh = nullptr /* x as bf16 (conv_forward wrote it so: *y_bf16) */);
void bn_relu_backward(hipStream_t s, const float* x, const float* y, const float* dA, const float* w, const float* save,
                      float* dx, float* dw, float* db, void* scratch, int64_t rows, int C, int tb_rows, bf16_t* dxb = nullptr,
                      const bf16_t* yb = nullptr, float* conv_dbias = nullptr, float* partial = nullptr, const BnSync* sync = nullptr,
                      ColsumJobs* defer = nullptr, const bf16_t* xh = nullptr /* x as bf16 */,
                      const bf16_t* dAh = nullptr /* d A as bf16 instead of fp32 (bn_partial4 path only) */,
                      int sums_chunks = 0 /* > 0: scratch already holds that many chunks of (sum d, sum d xhat) (conv_backward_data's bnb_chunks) */);
// yb: bf16 shadow of y (ReLU mask source); conv_dbias + partial (>= 4096*256 floats): fused bias gradient of the preceding conv, dx may then be null
// ctx_div: rows r share context row r / ctx_div (beam search keeps one context per image, model.lua:373)
void attention_forward(hipStream_t s, const float* ctx, const float* q, float* a, float* c, int64_t ldc, int B, int T, int Hd,
                       int ctx_div = 1, bf16_t* cb = nullptr, int64_t ldcb = 0, const bf16_t* ctxb = nullptr);
void gates_elem_bwd(hipStream_t s, const EpGatesBwd& ep, int M, int H);       // LSTM cell backward without a product (ops_gemm.hip)
bool attention_dual_ok(int T, int Hd, const bf16_t* ctxb, const bf16_t* ctxab);     // scores against the pre-multiplied context (round 6, ops_misc.hip: attn_bf16_kernel<..., DUAL>)
void attention_forward_dual(hipStream_t s, const float* h_top, int64_t ldh, float* a, float* c, int64_t ldc, int B, int T, int ctx_div, bf16_t* cb, int64_t ldcb,
                            const bf16_t* ctxb, const bf16_t* ctxab);
void attention_backward_dual(hipStream_t s, const float* a, const float* dc, int64_t lddc, float* ds, float* dq, bf16_t* dqb, float* dh_attn, int B, int T,
                             const bf16_t* ctxb, const bf16_t* ctxab);
void attention_backward(hipStream_t s, const float* ctx, const float* q, const float* a, const float* dc, int64_t lddc,
                        float* ds, float* dq, int B, int T, int Hd, bf16_t* dqb = nullptr, const bf16_t* ctxb = nullptr,
                        const float* cfwd = nullptr, int64_t ldcf = 0);      // cfwd (optional): the forward pass's weighted context of the same rows -- the streamed kernel then makes ONE pass (ops_misc.hip)
// ctxb: bf16 shadow of ctx, read instead of ctx by the register-resident kernels (T <= 64, Hd in {256, 512})
// d(ctx)[b,t,:] = sum_l a[l,b,t]*dc[l,b,:] + ds[l,b,t]*q[l,b,:]   (dc row stride lddc)
void attention_dctx(hipStream_t s, const float* a_all, const float* ds_all, const float* dc_all, int64_t lddc,
                    const float* q_all, float* dctx, int L, int B, int T, int Hd);
void logsoftmax_nll(hipStream_t s, const float* logits, int64_t ld, const int32_t* tgt, int64_t tgt_stride_t,
                    int64_t tgt_stride_b, int Bt, float* logp, float* dlogits, float* nll_rows, int64_t rows, int V,
                    float grad_scale);
void sum_to_scalar(hipStream_t s, const float* x, int64_t n, float* out);                 // out[0] = sum x
void gold_scores(hipStream_t s, const float* nll_rows, float* gold, int L, int B);       // gold[b] = -sum_t nll[t,b]
struct ZeroList { void* p[16]; size_t bytes[16]; int n = 0;
  void add(void* q, size_t b) { if (q && b && n < 16) { p[n] = q; bytes[n] = b; ++n; } } };
void zero_many(hipStream_t s, const ZeroList& z);    // all listed regions (16-byte aligned) in one launch
void colsum_accum(hipStream_t s, const float* A, int64_t ld, int64_t rows, int N, float* out, float* out2 = nullptr);   // out[n] (and out2[n]) += sum_r A[r][n]
// deferred form: up to 8 column sums finished by ONE launch (each small sum is otherwise its own ~10 us dispatch on the step's critical path)
struct ColsumJob { const float* A; int64_t ld, rows; int N; float* out; float* out2; int first, nb, chunks; };
struct ColsumJobs { int n, total; ColsumJob j[8]; };
void colsum_defer(ColsumJobs& g, const float* A, int64_t ld, int64_t rows, int N, float* out, float* out2 = nullptr);
void colsum_flush(hipStream_t s, ColsumJobs& g);
void embedding_gather(hipStream_t s, const float* table, const int32_t* tok, int64_t stride_t, int64_t stride_b, float* out,
                      int L, int B, int E);
// S [V][ncols] = rows of dz summed by their token (zeroed here; index: segsum_index_ints(rows, V) ints of scratch); then db1 / db2 += the column
// totals, dlookup [V][E] += S W[:, :E] (W [ncols][ldw]) and dW [ncols][ldw] (first E columns) += S^T lookup
bool segsum_supported(int ncols, int V, int E);
size_t segsum_index_ints(int rows, int V);
void segsum_by_token(hipStream_t s, const float* dz, int64_t ld, const int32_t* tok, int64_t st, int64_t sb, int L, int B, int ncols, int V, float* S, int* index,
                     float* db1, float* db2, const float* W, int64_t ldw, const float* lookup, int E, float* dlookup, float* dW);
void embedding_scatter_accum(hipStream_t s, const float* demb, const int32_t* tok, int64_t stride_t, int64_t stride_b,
                             float* dtable, int L, int B, int E, int V);
void dpre_tanh(hipStream_t s, const float* g1, const float* g2, const float* out, float* dpre, int64_t n, bf16_t* dpreb = nullptr, const DropSpec* drop = nullptr);
void pointwise(hipStream_t s, int op, const float* a, const float* b, float* y, int64_t n);   // AOCR_PW_* of include/aocr.h
void u8_to_f32(hipStream_t s, const uint8_t* src, float* dst, int64_t n);
void dropout_apply(hipStream_t s, const float* src, float* dst, bf16_t* dstb, int64_t n, const DropSpec& drop);
void dropout_apply_b(hipStream_t s, const bf16_t* src, float* dst, bf16_t* dstb, int64_t n, const DropSpec& drop);  // (g1+g2)*(1-out^2)
void copy2d_bf16(hipStream_t s, const float* src, int64_t lds, bf16_t* dst, int64_t ldd, int rows, int cols);
void copy2d(hipStream_t s, const float* src, int64_t lds, float* dst, int64_t ldd, int rows, int cols);
void copy2d_pair(hipStream_t s, const float* s0, const float* s1, int64_t lds, float* d0, float* d1, int64_t ldd, int rows, int cols);
struct DecInitArgs { float* c0[4]; float* h0[4]; bf16_t* hb[4]; float* feed0; bf16_t* outb; const float *cfw, *cbw, *hfw, *hbw; int B, He, Hd, Ld, copy_h; };
void dec_init(hipStream_t s, const DecInitArgs& a);              // the decoder's initial state in one launch (ops_misc.hip)
void sgd_clip_update(hipStream_t s, float* params, float* grads, const int64_t* group_off /*6 host values*/, float lr,
                     float clip, float* norms_out, void* scratch, int* err = nullptr,
                     float* bn_state = nullptr, const float* bn_snap = nullptr, int bn_n = 0);   // err = the time-out word block (cl_err): err[0] != 0: no update, bn_state <- bn_snap (the step's move of the running statistics is taken back), and the code moves to err[CL_ERR_STICKY]
constexpr int CL_ERR_LATCH = 12, CL_ERR_STICKY = 13;              // words of the cl_err block (0: step in flight, 1..4 encoder diagnostics, 8..11 exchange scratch, 16..: trash slots)
void step_snapshot(hipStream_t s, const float* bn_state, float* bn_snap, int n, int* err);        // start of a training step (ops_misc.hip)
size_t sgd_scratch_bytes();
void adadelta_update(hipStream_t s, float* params, float* grads, float* var, float* acc, int64_t n, float rho, float eps, float wd, int* err = nullptr,
                     float* bn_state = nullptr, const float* bn_snap = nullptr, int bn_n = 0);
// one 2-D piece of the per-step bf16 weight shadow refresh (shadow_jobs_kernel); tile0 = first 32x32 tile of the piece, tx = tiles per row
struct ShadowJob { const float* w; bf16_t* wb; bf16_t* wtb; int64_t ld, ldb, ldt; int R, C, tile0, tx; };
void shadow_jobs(hipStream_t s, const ShadowJob* jobs_dev, int njobs, int total_tiles, int tile_off = 0);      // tiles [tile_off, tile_off + total_tiles) of the table
// flat dictionary trie + the per-beam node ids of one decode step (mask == nullptr: unconstrained); needs V <= 64
struct TrieView { const unsigned long long* mask; const int32_t* base; const int32_t* child; const int32_t* loc_in; int32_t* loc_out; };
void beam_select(hipStream_t s, const float* logp, const int32_t* prev_tok, float* beam_scores, int32_t* tokens,
                 int32_t* parents, int B, int kin, int kout, int V, const float* logits = nullptr, int64_t ldl = 0,
                 const TrieView* tv = nullptr);
// logits != nullptr (and V <= 64): raw projector outputs, the LogSoftMax is applied inside (logp is then unused)
// projector + LogSoftMax + beam bookkeeping of one decode step in one launch (V <= 64, Hd % 4 == 0)
void project_select(hipStream_t s, const float* h, int64_t ldh, const float* wo, const float* bo, int Hd, const int32_t* prev_tok,
                    float* beam_scores, int32_t* tokens, int32_t* parents, int B, int kin, int kout, int V,
                    const TrieView* tv = nullptr);
void token_rows(hipStream_t s, const float* table, const int32_t* tok, int64_t stride, float* dst, int R, int width);   // dst[r] = table[tok[r*stride]-1]
// dst[b*kout+i][:] = src[(kin==1 ? b : b*kin + parents[b*kout+i])][:]
void gather_beam_rows_many(hipStream_t s, int n, const float* const* src, float* const* dst, int64_t ld, const int32_t* parents, int B, int kin, int kout, int width,
                           bf16_t* const* dstb = nullptr);      // dstb[i] (optional): bf16 copy of gathered tensor i
void gather_beam_rows(hipStream_t s, const float* src, int64_t lds, float* dst, int64_t ldd, const int32_t* parents, int B,
                      int kin, int kout, int width);
void beam_backtrace(hipStream_t s, const int32_t* hist_tok, const int32_t* hist_par, const float* beam_scores,
                    int32_t* labels, float* scores, int Lt, int B, int k);
void fill_i32(hipStream_t s, int32_t* p, int32_t v, int64_t n);
// Levenshtein distance between two id rows cut at the first EOS (utils.lua:55-94 over the strings of utils.lua:136-168)
void edit_distance(hipStream_t s, const int32_t* labels, const int32_t* targets, int B, int L, int32_t* dist, int32_t* target_len);

// ---- whole-sequence encoder recurrence (rnn_seq.hip): one workgroup owns 16 batch rows of one direction for all T steps
struct EncSeqDir {
  const bf16_t* w;         // recurrent weight, bf16 [4He][He]
  const float* zx;         // pre-computed input part incl. both biases, fp32 [T][B][4He]
  float* hs; float* cs;    // state slots [(T+2)][B][He]: slot t+1 holds step t
  bf16_t* hsb;             // bf16 shadow of hs
  float* gates;            // saved post-activation gates [T][B][4He]
  float* ctx;              // context + dir*He (element (b,t,j) at ctx[(b*T + t)*Hd + j]) or nullptr below the top layer
  int reverse;             // 1: the direction that walks t = T-1 .. 0
};
struct EncSeqFwdArgs { EncSeqDir d[2]; int B, T, He, Hd; int abl = 0; unsigned long long* dbg = nullptr; };
struct EncSeqBwdDir {
  const bf16_t* wt;        // transposed recurrent weight, bf16 [He][4He]
  const float* dh1;        // d h(t) from above: element (row, t, j) at dh1[row*dh1_row + t*dh1_t + j]
  int64_t dh1_row, dh1_t;
  const float* dh2;        // extra d h for the FIRST processed step (decoder initial state), row stride dh2_row; or nullptr
  int64_t dh2_row;
  float* dc;               // [B][He]: in = d c entering the first processed step, out = d c leaving the last one
  const float* dc_in = nullptr; int64_t dc_in_row = 0;   // cluster kernels, first chunk: d c entering the first step read from here (row stride dc_in_row) instead of dc -- the decoder's [B][2 He] initial-state gradient without a split copy
  const float* gates;      // saved gates [T][B][4He]
  const float* cs;         // cell-state slots [(T+2)][B][He]
  float* dz; bf16_t* dzb;  // out: d z [T][B][4He], fp32 and bf16
  float* dbi = nullptr; float* dbh = nullptr;   // cluster kernels only: both bias gradients (+= sum of d z over rows and steps; no fp32 d z is written)
  int forward_dir;         // 1: the direction whose forward pass walked t = 0..T-1 (its BPTT walks T-1..0, c_prev = slot t)
};
struct EncSeqBwdArgs { EncSeqBwdDir d[2]; int B, T, He; };
bool enc_seq_supported(int B, int He, int blocks_limit);
// ---- the same recurrences on clusters of CUs with register-resident weights (rnn_cluster.hip)
struct EncClFwdArgs { EncSeqDir d[2]; int B, T, He, Hd, groups; unsigned epoch; unsigned long long* xbuf; int* err; int gid0 = 0, ngid = 0; unsigned long long* xtab = nullptr; int force_remote = 0; int it0 = 0, it1 = 0 /* iterations [it0, it1) of the T in this launch; it1 = 0: all */, gslot = 0 /* group-slot offset into xbuf / xtab (one set of slots per layer) */; int rh = 16 /* batch rows per 16-column tile: 16, or 8 (half tiles; set by enc_cluster_forward) */; };
struct EncClBwdArgs { EncSeqBwdDir d[2]; int B, T, He, groups; unsigned epoch; unsigned long long* pbuf; int* err; int gid0 = 0, ngid = 0; unsigned long long* xtab = nullptr; int force_remote = 0; int it0 = 0, it1 = 0, gslot = 0; int rh = 16 /* as EncClFwdArgs::rh; set by enc_cluster_backward */; };
bool enc_cluster_plan(int B, int He, int T, int cus, int& G, int& RT, int& groups);
size_t enc_cluster_xbuf_bytes(int B, int He);
size_t enc_cluster_pbuf_bytes(int B, int He);
void enc_cluster_forward(hipStream_t s, const EncClFwdArgs& a, int G, int RT, int reserve_cus = 0, int concurrent = 1);      // concurrent: launches of this size that run side by side (layer wavefront): half tiles only if they all fit
void enc_cluster_backward(hipStream_t s, const EncClBwdArgs& a, int G, int RT, int reserve_cus = 0, int concurrent = 1);
// teacher-forced decoder loop in one launch (dec_cluster.hip): Hd = 512, two layers, input feed, bf16 mode
struct DecClFwdArgs {
  int B, T, L; unsigned epoch; int group0 = 0, ngroups = 0, force_remote = 0; int no_early = 0;   /* greedy decode: do not leave the loop when every row of a group has finished (debugging aid) */
  const bf16_t *w1i, *w1h, *w2i, *w2h, *wc;           // bf16 shadows: [4 Hd][Hd] x 4, W_c [Hd][2 Hd]
  const float *b2i, *b2h;                              // layer-2 biases (layer 1's are inside zx1)
  const float* zx1;                                    // [L][B][4 Hd]: embedding part of layer 1 + both biases
  const bf16_t *ctxb, *ctxa;                           // [B][T][Hd]: encoder context and context . W_a
  float* cs[2]; bf16_t* hsb[2];                        // [L + 1][B][Hd], slot 0 = initial state (the fp32 h is not written: every reader takes the bf16 copy)
  float* gates[2];                                     // [L][B][Hd][4] post-activation (i, f, o, g INTERLEAVED per unit), or nullptr
  float *a_all, *out; bf16_t *cat_b, *out_b;           // [c ; h_top] only as its bf16 copy
  unsigned long long *xbuf, *xtab; int* err; unsigned long long* stamps = nullptr;
  // greedy decode (the kernel's DEC variant): zx1 is the per-token table [V][4 Hd]; tok0 = the GO tokens (row stride tok0_stride = the label width)
  const int32_t* tok0 = nullptr; int tok0_stride = 0; const float *wo = nullptr, *bo = nullptr; int V = 0;
  float* pbuf = nullptr; unsigned* tokx = nullptr; int32_t* labels = nullptr; float* scores = nullptr; int pgroups = 0;   /* dec_chain.hip: groups of the whole batch (the partial logits are kept per step parity) */
  const unsigned long long* trie_mask = nullptr; const int32_t* trie_base = nullptr; const int32_t* trie_child = nullptr;   // -use_dictionary (flat trie of include/aocr.h) or null
  // beam search on the chain kernel (dec_chain.hip, BEAM variant): k hypotheses per image, history [L][B][k] + final scores [B][k] for beam_backtrace
  int beam = 0, rows_slot = 0; int32_t *hist_tok = nullptr, *hist_par = nullptr; float* beam_scores = nullptr;
  // nn.Dropout(p) (training, LSTM.lua:68-69,116-118): masks of layer 2's input (site 2) and of the attention output (site 16), flat index
  // = step * B * Hd + row * Hd + unit added to .off = 0; hm_b [L][B][Hd]: the masked bf16 copy of h1 (operand of layer 2 and of its weight gradient)
  DropSpec drop_h, drop_out; bf16_t* hm_b = nullptr;
  // round 4, teacher-forced loop: zx1 as the per-token table [V][4 Hd] too (zx_tok = the input tokens, token of (step t, row b) = zx_tok[t zx_st + b zx_sb]):
  // no (L B, 4 Hd) gate-input tensor is written or read; V must be set
  const int32_t* zx_tok = nullptr; int64_t zx_st = 0, zx_sb = 0;
};
// decoder BPTT in one launch (dec_cluster.hip); reads what the forward cluster kernel saved (interleaved gates)
struct DecClBwdArgs {
  int B, T, L; unsigned epoch; int group0 = 0, ngroups = 0, force_remote = 0;
  const bf16_t *w2i_t, *w2h_t, *w1h_t, *w1f_t;          // transposed bf16 shadows [Hd][4 Hd]: W2_i2h, W2_h2h, W1_h2h, W1_i2h[:, E:]
  const bf16_t *wc_t, *wa_t;                           // [2 Hd][Hd], [Hd][Hd]
  const float *dout_proj, *out;                        // [L][B][Hd] projector gradient; [L + 1][B][Hd] attention outputs
  const float* a_all; const bf16_t* ctxb;              // [L][B][T]; [B][T][Hd]
  const float* cs[2]; const float* gates[2];           // [L + 1][B][Hd]; [L][B][Hd][4] (interleaved)
  float* dpre; bf16_t* dpre_b;                         // [L][B][Hd]
  float* dcat;                                         // [L][B][2 Hd]: only the c half is written
  float* ds_all; float* dq; bf16_t* dq_b;              // [L][B][T]; [L][B][Hd]
  float* dz[2]; bf16_t* dzb[2];                        // [L][B][4 Hd]
  float* dc_st[2]; float* dh_rec[2]; float* dfeed;     // [B][Hd]: gradients of the initial state
  unsigned long long *xbuf, *xtab; int* err; unsigned long long* stamps = nullptr;
  DropSpec drop_h, drop_out;                           // as in DecClFwdArgs (the forward kernel stored the MASKED attention output)
};
// beam search (2 <= beam <= 8, V <= 40) as launches of dec_chain.hip's kernel: a.epoch = the first of dec_chain_beam_passes() consecutive epochs,
// a.pbuf / a.xtab sized for dec_chain_beam_group_cap() groups; four rolling step slots inside the [L + 1][B][Hd] buffers bound the groups of a launch (..._supported: at least one fits)
bool dec_chain_beam_supported(int B, int L, int beam, int V);
int dec_chain_beam_passes(int B, int L, int beam);
int dec_chain_beam_group_cap();
void dec_chain_beam_forward(hipStream_t s, const DecClFwdArgs& a);
bool dec_cluster_supported(int Hd, int Ld, int input_feed, int T, int L, int cus);
bool dec_cluster_bwd_supported(int Hd, int Ld, int input_feed, int T, int L, int cus);
size_t dec_cluster_bwd_xbuf_bytes(int B);
void dec_cluster_backward(hipStream_t s, const DecClBwdArgs& a);
size_t dec_cluster_xbuf_bytes(int B);
size_t dec_cluster_xtab_bytes(int B);
void dec_cluster_forward(hipStream_t s, const DecClFwdArgs& a, bool greedy_decode = false);
size_t dec_cluster_pbuf_bytes(int B);
void enc_seq_backward(hipStream_t s, const EncSeqBwdArgs& a);
void enc_seq_forward(hipStream_t s, const EncSeqFwdArgs& a);
// data path (data.hip): 255*rgb2y + image.scale to (out_h, out_w) for n images sharing out_w
void preprocess_lines(hipStream_t s, const uint8_t* src, const aocr_image_desc* desc, int n_images, int out_h, int out_w, float* out);
}  // namespace aocr
