/* aocr.h -- C ABI of libaocr: the MI355X (gfx950) implementation of the
 * CNN -> BiLSTM -> attention-decoder train / decode step of
 * da03/torch-Attention-OCR.
 *
 * The reference has no FFI of its own: its hot path runs inside un-vendored
 * Torch7 packages that src/train.lua:4-9 and src/model/model.lua:3-7 only
 * `require`.  The entry points below are what a LuaJIT `ffi.cdef` (see
 * INTEGRATION.md) or a ctypes binding (torch-attention-ocr_amd/aocr/_lib.py)
 * binds instead; each one cites the reference code it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from aocr_last_error() (thread-local).  Nothing throws.
 *   - all pointers named *_dev are DEVICE pointers (HBM) owned by the caller;
 *     the library never allocates device memory: parameters, gradients and the
 *     workspace arena are handed in at aocr_model_create().
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only enqueues work on it and never synchronises.
 *   - activations are fp32, channels-last (B,H,W,C); token ids are 1-based int32
 *     (1 PAD, 2 GO, 3 EOS: src/train.lua:53).
 *   - the ABI is not re-entrant per aocr_model (Lua is single threaded).
 */
#ifndef AOCR_H
#define AOCR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOCR_VERSION 1
#define AOCR_NUM_GROUPS 5           /* cnn, enc_fw, enc_bw, decoder, projector: model.lua:150 */
#define AOCR_COMPUTE_F32 0          /* v_mfma_f32_32x32x2_f32, exact fp32 */
#define AOCR_COMPUTE_BF16 1         /* bf16 operands, fp32 accumulate */

typedef struct aocr_model aocr_model;

/* Hyper-parameters: src/train.lua:41-62, src/model/model.lua:83-96. */
typedef struct aocr_config {
  int32_t batch_size;       /* max rows per step                        (-batch_size) */
  int32_t img_h;            /* 32, src/data/data_gen.lua:16 */
  int32_t max_img_w;        /* widest crop the workspace is sized for */
  int32_t enc_hidden;       /* -encoder_num_hidden */
  int32_t enc_layers;       /* -encoder_num_layers */
  int32_t dec_layers;       /* -decoder_num_layers */
  int32_t vocab;            /* -target_vocab_size (39) */
  int32_t emb;              /* -target_embedding_size (20) */
  int32_t input_feed;       /* -input_feed */
  int32_t max_decoder_l;    /* -max_decoder_l (50) */
  int32_t max_beam;         /* largest -beam_size the workspace is sized for */
  int32_t compute;          /* AOCR_COMPUTE_* */
} aocr_config;

const char* aocr_last_error(void);
int aocr_version(void);

/* ---- parameter layout (replaces nn.Module:getParameters(), model.lua:163-168)
 * One flat fp32 buffer holds the 5 groups back to back; within a group the order
 * is Torch7's (module order, weight then bias).  Conv weights are stored
 * [Cout][kH][kW][Cin] (channels-last taps); everything else as in Torch7. */
int aocr_param_counts(const aocr_config* cfg, int64_t counts[AOCR_NUM_GROUPS]);
/* Enumerate named tensors: returns 0 and fills the outputs for index < n, 1 past the end.
 * offset is relative to the start of the flat buffer (not of the group). */
int aocr_param_entry(const aocr_config* cfg, int32_t index, char name[64], int32_t* group,
                     int64_t* offset, int32_t* ndim, int64_t shape[4]);
/* BatchNorm running statistics (not parameters): [rm3(256) rv3 rm5(512) rv5 rm7(512) rv7]. */
int64_t aocr_bn_state_count(void);

size_t aocr_workspace_bytes(const aocr_config* cfg);

/* ---- model handle (replaces Model:create/_build, model.lua:83-223) */
int aocr_model_create(const aocr_config* cfg, float* params_dev, float* grads_dev,
                      float* bn_state_dev, void* workspace_dev, size_t workspace_bytes,
                      void* stream, aocr_model** out);
int aocr_model_destroy(aocr_model* m);
int aocr_model_set_stream(aocr_model* m, void* stream);

/* ---- fused sequence-level entry points (what Model:step uses) */

/* Health of the whole-sequence ("cluster") kernels: the compute units that share a block of batch rows wait for each other with BOUNDED
 * spins; if a wait ever times out (it cannot unless another kernel keeps part of the chip busy for ~0.3 s) the kernel records a code,
 * finishes, and the step's results are invalid: aocr_sgd_step / aocr_adadelta_step then leave the parameters untouched (device-side
 * predicate on the same flag, no host sync).  The optimizer call CONSUMES the code (round 6): it moves it to a second word that waits for
 * the host, so exactly the step that timed out is skipped -- a host that polls this call only every N steps loses that one step, never its
 * weights and never the healthy steps behind it (before round 6 every step up to the poll was dropped).  A code raised by a call that has
 * no optimizer behind it (aocr_decode*) is moved aside the same way by the next aocr_train_forward_backward, so it cannot cancel that step.
 * *code = 0: healthy; otherwise the most recent code not yet reported; cleared by the call (read and clear).
 * Under data parallelism the code is a GLOBAL decision (round 4): aocr_allreduce_grads sums a time-out flag with the exchange, and a rank whose own
 * kernels were healthy while a peer's were not reads 0x7e -- so every rank's optimizer skips the update and every host repeats the step together.
 * The optimizer call that skips an update also restores the BatchNorm running statistics to their values at the start of that step (device side, round 5):
 * a skipped step leaves NO trace in the model, so the host may repeat the batch at once, later (whenever it polls) or never.
 * A caller that sums the gradient buckets itself (aocr_stream_wait_grads) must also take the MAX over ranks of the status word (tap "cl_err"[0]) before
 * aocr_sgd_step, as aocr_allreduce_grads does inside the library; the Python mirror does (Model._exchange_timeout_flag).
 * Synchronises the model's stream.  No reference counterpart.
 * These kernels need the device to themselves: a launch occupies every compute unit, so the steps of two models (or two processes)
 * must not run concurrently on ONE device -- AOCR_NO_CLUSTER=1 AOCR_NO_DEC_CLUSTER=1 selects the per-step launch chains for that case. */
int aocr_cluster_status(aocr_model* m, int32_t* code);

/* nn.Dropout(p) of LSTM.lua:68-69 (input of every LSTM layer above the first, encoder and decoder) and :116-118 (attention output),
 * active in the training step only (model.lua:284 `training()`; decode / forward_only run `evaluate()`).  The mask is a counter-based
 * function of (seed, train_step, site, element index) -- splitmix64, restated in oracle/oracle_torch.py::dropout_mask -- so a step is
 * reproducible and the oracle can replay it; kept activations are scaled by 1/(1-p) (nn.Dropout v2).  Call before every training
 * step with the step counter (the reference draws from torch's global generator instead: the mask VALUES cannot match, the
 * distribution does).  p = 0 (default) disables it.  (Round 3: the whole-sequence decoder kernels evaluate the same masks.) */
int aocr_set_dropout(aocr_model* m, double p, uint64_t seed, uint64_t train_step);

/* feval of model.lua:284-696 with forward_only=false: CNN forward (training-mode
 * BatchNorm, running stats updated), encoder fw/bw loops, teacher-forced decoder
 * loop, loss, hand-ordered BPTT, CNN backward.  Gradients are ZEROED then
 * accumulated into grads_dev with d(loss) scaled by grad_scale (= 1/batch_size in
 * the reference, model.lua:645-647; 1/global_batch under data parallelism).
 * loss_dev[0] = sum of NLL over the step (un-scaled, = loss*batch_size of :701).
 * images_dev (B,1,32,W) fp32 values 0..255; targets/targets_eval (B,L) int32
 * (src/data/data_gen.lua:107-117). */
int aocr_train_forward_backward(aocr_model* m, const float* images_dev, const int32_t* targets_dev,
                                const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L,
                                float grad_scale, float* loss_dev);

/* Data-parallel overlap: the flat gradient vector completes back to front
 * during the backward pass -- bucket 0 = decoder + projector groups, 1 = both encoder groups,
 * 2 = the CNN from conv5 upwards, 3 = conv1..conv4 -- and an event marks each point.
 * aocr_grad_buckets gives the [begin, end) float ranges of the buckets inside grads_dev;
 * aocr_stream_wait_grads makes `stream` wait until bucket `bucket` of the LAST enqueued
 * aocr_train_forward_backward is complete, so that its all-reduce can run on a second stream
 * beside the rest of the backward pass.  The caller joins that stream before aocr_sgd_step. */
#define AOCR_GRAD_BUCKETS 4
int aocr_grad_buckets(const aocr_config* cfg, int64_t begin[AOCR_GRAD_BUCKETS], int64_t end[AOCR_GRAD_BUCKETS]);
int aocr_stream_wait_grads(aocr_model* m, int32_t bucket, void* stream);

/* ---- data parallelism inside the library: one process per GPU, ONE exchange step -- the sum over ranks of the flat
 * gradient vector between feval and the per-group clip (optim_sgd.lua:38 -> :40) -- plus, with sync_bn, the per-channel BatchNorm
 * sums of the three BatchNorm layers (cnn.lua:23,32,41) in the forward and the backward pass, so that N ranks on slices of a batch
 * compute what one GPU computes on the whole batch (training-mode statistics included; running statistics stay identical on all ranks).
 * Provider 1: RCCL over xGMI.  rank 0 calls aocr_comm_unique_id, the host hands the 128 bytes to the other ranks (file, socket, MPI:
 * its choice), every rank calls aocr_comm_init_rank (collective, like ncclCommInitRank).  librccl.so is bound at run time.
 * Provider 2: a host callback that sums `count` elements (dtype & 0xff: 0 = fp32, 1 = fp64) of a device buffer over the ranks in place,
 * enqueued on `stream` (the Python mirror routes torch.distributed through it; the tests use gloo).  dtype & AOCR_COMM_CHANNEL_BN
 * marks the BatchNorm sums: they are issued from inside the forward / backward pass while the gradient buckets are issued after it,
 * and a communicator runs its operations in issue order -- give that channel a communicator (process group) of its own, as provider 1
 * does (ncclCommSplit), or bucket 0 queues behind the last BatchNorm-backward sum and the overlap with the backward pass is lost.
 * aocr_allreduce_grads: after aocr_train_forward_backward and before aocr_sgd_step; the buckets of aocr_grad_buckets are summed on a
 * second stream as the backward pass completes them (overlap), loss_dev (optional, 1 float) is summed too, and the model's stream
 * waits for the last bucket.  Pass grad_scale = 1 / GLOBAL batch to aocr_train_forward_backward.
 * Exchange policy (round 4): no collective is in flight while a whole-sequence kernel runs -- with a communicator attached bucket 0 (decoder + projector)
 * is released behind the encoder BPTT (AOCR_COMM_EARLY_BUCKET0=1: as soon as it is complete; the encoder kernels then leave AOCR_COMM_RESERVE_CUS,
 * default 32, compute units free).  aocr_comm_init_rank fails if ncclCommSplit fails (AOCR_ONE_COMM=1 on every rank shares one communicator). */
#define AOCR_COMM_CHANNEL_BN 0x100
typedef int (*aocr_allreduce_fn)(void* user, void* buf_dev, int64_t count, int32_t dtype, void* stream);
int aocr_comm_unique_id(char id[128]);
int aocr_comm_init_rank(aocr_model* m, const char id[128], int32_t nranks, int32_t rank, int32_t sync_bn);
int aocr_comm_set_callback(aocr_model* m, aocr_allreduce_fn fn, void* user, int32_t nranks, int32_t sync_bn);
int aocr_allreduce_grads(aocr_model* m, float* loss_dev);
int aocr_comm_destroy(aocr_model* m);
/* What is attached: *nranks (1 without a communicator), *sync_bn (0 / 1), *provider (0 none, 1 RCCL, 2 host callback).  Any pointer may be NULL. */
int aocr_comm_info(aocr_model* m, int32_t* nranks, int32_t* sync_bn, int32_t* provider);
/* Measurement: milliseconds the model's stream waited for the exchange stream at the end of the LAST aocr_allreduce_grads -- the part
 * of the gradient exchange that the backward pass did not hide (0 without a communicator).  Synchronises that step. */
int aocr_comm_exposed_ms(aocr_model* m, float* ms);

/* optim.sgd_list, src/optim/optim_sgd.lua:38-95 with the options the reference
 * leaves at 0: per group, if ||g||_2 > clip then g *= clip/||g||_2; w -= lr*g.
 * norms_dev (optional, 2*5 floats): {param norm, grad norm} per group, as printed
 * by optim_sgd.lua:49.  A data-parallel all-reduce of grads_dev goes before this call. */
int aocr_sgd_step(aocr_model* m, float lr, float clip, float* norms_dev);

/* optim.adadelta_list, src/optim/optim_adadelta.lua:19-62 (the optimizer the reference ships
 * next to sgd_list), one fused pass over all parameters: var = rho*var + (1-rho)*g^2;
 * delta = sqrt(acc+eps)/sqrt(var+eps)*g; w -= delta; acc = rho*acc + (1-rho)*delta^2.
 * state_dev: 2*n floats {paramVariance | accDelta} (n = sum of aocr_param_counts), zeroed by
 * the caller before the first step and kept between steps.  weight_decay: g += wd*w first
 * (what optim_adadelta.lua:37 means; the line itself would raise).  No clipping (:31-57). */
int aocr_adadelta_step(aocr_model* m, float rho, float eps, float weight_decay, float* state_dev);

/* Teacher-forced forward only (no gradients).  training!=0 uses batch statistics
 * in BatchNorm (without touching running stats); logits_dev (L,B,vocab) receives the
 * pre-LogSoftMax projector output (output_projector.lua:5), loss_dev[0] the NLL sum. */
int aocr_forward_logits(aocr_model* m, const float* images_dev, const int32_t* targets_dev,
                        const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L,
                        int32_t training, float* logits_dev, float* loss_dev);

/* feval with forward_only=true (model.lua:321-536, 570-627): eval-mode CNN, encoder,
 * beam search over max_decoder_l steps (beam 1 = greedy), back-trace, then the
 * teacher-forced gold pass.  labels_dev (B,max_decoder_l) int32, scores_dev (B),
 * gold_scores_dev (B), loss_dev[0] = gold-pass NLL sum.  No dictionary (trie == nil). */
int aocr_decode(aocr_model* m, const float* images_dev, const int32_t* targets_dev,
                const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, int32_t beam,
                int32_t* labels_dev, float* scores_dev, float* gold_scores_dev, float* loss_dev);

/* The dictionary trie of loadDictionary (utils.lua:177-218) in flat, device-resident form.
 * Node 0 is the start symbol's node (trie[2]).  Node n has a child for vocab id v (1-based,
 * v <= 64) iff bit v-1 of child_mask_dev[n] is set; the children of n are
 * child_dev[child_base_dev[n] ...] in ascending v.  A node reached through EOS (3) is an
 * ordinary node (childless unless -allow_digit_prefix made it the root, utils.lua:193-198). */
typedef struct aocr_trie {
  const uint64_t* child_mask_dev;   /* [n_nodes] */
  const int32_t* child_base_dev;    /* [n_nodes] */
  const int32_t* child_dev;         /* [n_edges] */
  int32_t n_nodes, n_edges;
} aocr_trie;

/* aocr_decode with -use_dictionary (model.lua:380-387,405-445,460-513): at every step only
 * candidates whose token continues the beam's trie node are admissible (after the first step
 * PAD always is, :469); when fewer than `beam` candidates are admissible the best admissible one
 * fills the remaining beams (:419-433; the same rule replaces the broken fallback of :477-497).
 * trie == NULL is aocr_decode.  Needs target_vocab_size <= 64. */
int aocr_decode_dict(aocr_model* m, const float* images_dev, const int32_t* targets_dev,
                     const int32_t* targets_eval_dev, int32_t B, int32_t W, int32_t L, int32_t beam,
                     const aocr_trie* trie, int32_t* labels_dev, float* scores_dev,
                     float* gold_scores_dev, float* loss_dev);

/* Debug / parity taps: device pointer + shape of a named intermediate of the last
 * step ("feats" (T,B,512), "context" (B,T,2He), "logits" (L,B,40), "dfeats", "dcontext"). */
int aocr_get_tensor(aocr_model* m, const char* name, const void** ptr_dev, int32_t* ndim, int64_t shape[4]);

/* Times `iters` launches of one hot kernel of the LAST step's shape with HIP events
 * on the model's stream; which: 0 = conv6 forward implicit GEMM (largest layer), 1 = conv6 filter gradient (the split-K
 * kernel + the sum of its slabs, as the backward pass launches them; the result goes to scratch).
 * ms_per_launch and flops_per_launch are host outputs (this call synchronises).
 * which >= 2 (round 4): the BANDWIDTH-bound kernels of the bf16 training step, each replayed with the arguments and on the buffers of the last
 * aocr_train_forward_backward (whose image buffer must still be valid); *flops_per_launch then returns the ALGORITHMIC BYTES of one launch
 * (what the kernel must read and write once), so bytes / ms = the HBM rate to hold against ~6.3 TB/s achievable.
 * These replays overwrite activations / gradient maps / gradients: taps and gradients are undefined until the next step. */
#define AOCR_PK_CONV6_FWD 0
#define AOCR_PK_CONV6_WGRAD 1
#define AOCR_PK_CONV1_FWD 2      /* normalise + conv1 + ReLU + 2x2 pool (cnn.lua:9-15): image in, pooled bf16 map out */
#define AOCR_PK_CONV1_BWD 3      /* its filter / bias gradient (window recomputed; no d(image)) */
#define AOCR_PK_BN_FWD 4         /* conv5's BatchNorm + ReLU (cnn.lua:32-33): finalize + apply pass (the sums come from the conv epilogue) */
#define AOCR_PK_BN_BWD 5         /* its backward: sums pass + apply pass */
#define AOCR_PK_UNPOOL 6         /* conv6's (2,1) un-pool + ReLU backward (cnn.lua:37-38) */
#define AOCR_PK_ATTN_DCTX 7      /* d(context) summed over the L decoder steps (model.lua:652-653) */
#define AOCR_PK_SPLITK 8         /* sum of conv6's split-K filter-gradient slabs */
#define AOCR_PK_LAST 8
int aocr_profile_kernel(aocr_model* m, int32_t which, int32_t iters, float* ms_per_launch, double* flops_per_launch);

/* Per-family timing of the fused step with HIP events on the model's stream (measurement only).  After
 * aocr_profile_enable(m, 1) every fused entry point records an event wherever the work changes family; aocr_profile_read
 * synchronises the stream, returns the milliseconds accumulated per family since the last read (ms[AOCR_PROF_FAMILIES]) and the
 * number of marks, and clears the record.  Families: */
#define AOCR_PROF_OTHER 0        /* weight shadows, zero fills, copies, loss, embedding */
#define AOCR_PROF_CONV_FWD 1     /* conv2..conv7 forward implicit GEMMs (cnn.lua:17-41) */
#define AOCR_PROF_CONV_DGRAD 2   /* their data gradients */
#define AOCR_PROF_CONV_WGRAD 3   /* their filter gradients */
#define AOCR_PROF_BN 4           /* BatchNorm(+ReLU) forward and backward (cnn.lua:23,32,41) */
#define AOCR_PROF_POOL_CONV1 5   /* conv1 forward / backward and the un-pool passes */
#define AOCR_PROF_ENC_SEQ 6      /* the encoder recurrences, forward and BPTT (model.lua:294-316, 664-690) */
#define AOCR_PROF_RNN_GEMM 7     /* hoisted input projections, projector, every LSTM / attention weight gradient */
#define AOCR_PROF_DEC_FWD 8      /* decoder step chain forward (model.lua:553-568) */
#define AOCR_PROF_DEC_BWD 9      /* decoder step chain BPTT (model.lua:643-661) */
#define AOCR_PROF_SGD 10         /* clip + update (optim_sgd.lua:38-95) */
#define AOCR_PROF_DECODE 11      /* beam / greedy decode chain (model.lua:376-536) */
#define AOCR_PROF_FAMILIES 12
int aocr_profile_enable(aocr_model* m, int32_t on);
int aocr_profile_read(aocr_model* m, float ms[AOCR_PROF_FAMILIES], int32_t* marks);

/* ---- module-level entry points (the nn.Module surface of the files under src/model) ---- */

/* C[M,N] (ldc) = op(A) * op(B) (+bias[n]) ; a_kmajor: 1 = A stored [M][K] (lda), 0 = [K][M];
 * b_kmajor: 1 = B stored [N][K] (ldb), 0 = [K][N].  nn.Linear forward is (1,1) with bias
 * (LSTM.lua:79-88), its gradInput (1,0), its gradWeight (0,0).
 * accumulate: bit field -- 1: C += (instead of C =), 2: ReLU on the result, 4: tanh on the result (nn.Tanh fused behind
 * nn.LinearNoBias, LSTM.lua:155-157); 0 / 1 keep their round-1 meaning. */
#define AOCR_GEMM_ACCUMULATE 1
#define AOCR_GEMM_RELU 2
#define AOCR_GEMM_TANH 4
int aocr_gemm(void* stream, int32_t compute, const float* A_dev, int64_t lda, int32_t a_kmajor,
              const float* B_dev, int64_t ldb, int32_t b_kmajor, float* C_dev, int64_t ldc,
              int32_t M, int32_t N, int32_t K, const float* bias_dev, int32_t accumulate);

/* cudnn.SpatialConvolution + cudnn.ReLU + cudnn.SpatialMaxPooling (cnn.lua:12-42), fused.
 * x (B,H,W,Cin) channels-last; w [Cout][k][k][Cin]; stride 1.  pool: 0 none, 1 = 2x2/2, 2 = kH2 kW1 / (2,1).
 * relu applies before the pool.  y is (B,Ho',Wo',Cout); idx (same shape, uint8, may be NULL when pool=0)
 * records the arg-max position inside the window. */
int aocr_conv2d_forward(void* stream, int32_t compute, const float* x_dev, const float* w_dev, const float* bias_dev,
                        float* y_dev, uint8_t* idx_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                        int32_t ksize, int32_t pad, int32_t relu, int32_t pool);
/* gradInput / gradWeight+gradBias of the same layer w.r.t. the PRE-pool, PRE-relu output gradient dy (B,Ho,Wo,Cout). */
int aocr_conv2d_backward_data(void* stream, int32_t compute, const float* dy_dev, const float* w_dev, float* dx_dev,
                              int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad);
int aocr_conv2d_backward_filter(void* stream, int32_t compute, const float* x_dev, const float* dy_dev, float* dw_dev,
                                float* dbias_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                int32_t ksize, int32_t pad);
/* Routes d(pooled) back through max-pool + ReLU: dy (B,Ho,Wo,C) from dpooled, idx and pooled (>0 test). */
int aocr_unpool_relu_backward(void* stream, const float* dpooled_dev, const float* pooled_dev, const uint8_t* idx_dev,
                              float* dy_dev, int32_t B, int32_t Ho, int32_t Wo, int32_t C, int32_t pool);

/* First layer, cnn.lua:9-15 fused: (x-128)/128, conv 1->64 3x3 p1, ReLU, maxpool 2x2. x (B,32,W) -> y (B,16,W/2,64). */
int aocr_conv1_forward(void* stream, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                       int32_t B, int32_t H, int32_t W);
int aocr_conv1_backward(void* stream, const float* x_dev, const float* w_dev, const float* bias_dev,
                        const float* dy_pooled_dev, float* dw_dev, float* dbias_dev, int32_t B, int32_t H, int32_t W);

/* nn.SpatialBatchNormalization (+ the ReLU that follows it in cnn.lua:23-24,32-33,41-42).
 * x,y (rows,C) channels-last.  training: batch stats (biased var, eps 1e-5), running stats updated with
 * momentum 0.1 and unbiased var when update_running; save_dev gets [mean(C), invstd(C)].
 * tb_rows>0 writes y transposed from (B,T,C) to (T,B,C) with B = tb_rows (cnn.lua:44-45 + model.lua:288).
 * scratch_dev: AOCR_BN_SCRATCH_BYTES of device memory the call may overwrite (partial sums), shared between calls on one stream. */
#define AOCR_BN_SCRATCH_BYTES (8u << 20)
int aocr_batchnorm_relu_forward(void* stream, const float* x_dev, float* y_dev, const float* weight_dev,
                                const float* bias_dev, float* running_mean_dev, float* running_var_dev,
                                float* save_dev, void* scratch_dev, int64_t rows, int32_t C, int32_t training,
                                int32_t update_running, int32_t tb_rows);
/* dy_out = d/dx of relu(bn(x)) given dA (gradient at the ReLU output) and y (the ReLU output).  dweight/dbias accumulate. */
int aocr_batchnorm_relu_backward(void* stream, const float* x_dev, const float* y_dev, const float* dA_dev,
                                 const float* weight_dev, const float* save_dev, float* dx_dev, float* dweight_dev,
                                 float* dbias_dev, void* scratch_dev, int64_t rows, int32_t C, int32_t tb_rows);

/* One LSTM cell, LSTM.lua:79-105 (gate order in,forget,out,g; two biases): z = W_i2h x + b_i2h + W_h2h h_prev + b_h2h.
 * x (B,in), in and H multiples of 16.  Outputs c,h (B,H) and gates (B,4H) post-activation. */
int aocr_lstm_cell_forward(void* stream, int32_t compute, const float* x_dev, int32_t in_size, const float* h_prev_dev,
                           const float* c_prev_dev, const float* w_i2h_dev, const float* b_i2h_dev,
                           const float* w_h2h_dev, const float* b_h2h_dev, float* c_dev, float* h_dev,
                           float* gates_dev, int32_t B, int32_t H);
/* The same cell with the input part pre-computed, as the fused path hoists it out of the time loops (and for input widths that are
 * not multiples of 16: the decoder's first layer sees E + Hd = 532 columns): zx (B,4H) row stride ldzx = W_i2h x + b_i2h + b_h2h
 * (aocr_gemm with the summed bias), the call adds W_h2h h_prev and applies the gates. */
int aocr_lstm_cell_forward_zx(void* stream, int32_t compute, const float* zx_dev, int64_t ldzx, const float* h_prev_dev,
                              const float* c_prev_dev, const float* w_h2h_dev, float* c_dev, float* h_dev, float* gates_dev,
                              int32_t B, int32_t H);
/* Backward of the cell given d(c_out), d(h_out): writes dz (B,4H), dc_prev; dx/dh_prev via aocr_gemm(dz, W). */
int aocr_lstm_cell_backward(void* stream, const float* dc_dev, const float* dh_dev, const float* gates_dev,
                            const float* c_prev_dev, const float* c_dev, float* dz_dev, float* dc_prev_dev,
                            int32_t B, int32_t H);

/* create_decoder_attn, LSTM.lua:124-162, the score/softmax/context core given q = W_a h_top:
 * a = softmax_T(ctx . q), c = a . ctx.  ctx (B,T,Hd), q (B,Hd) -> a (B,T), c written at c_dev with row stride ldc. */
int aocr_attention_forward(void* stream, const float* ctx_dev, const float* q_dev, float* a_dev, float* c_dev,
                           int64_t ldc, int32_t B, int32_t T, int32_t Hd);
/* given dc (row stride lddc): ds (B,T) = softmax-backward of the scores, dq (B,Hd). d(ctx) is assembled over all
 * decoder steps by the fused path. */
int aocr_attention_backward(void* stream, const float* ctx_dev, const float* q_dev, const float* a_dev,
                            const float* dc_dev, int64_t lddc, float* ds_dev, float* dq_dev, int32_t B, int32_t T,
                            int32_t Hd);

/* Element-wise helpers of the module-level surface (n fp32 elements; y may alias a or b):
 *   AOCR_PW_ADD        y = a + b                  nn.CAddTable (LSTM.lua:88), gradient fan-in of a shared input
 *   AOCR_PW_TANH_BWD   y = a * (1 - b^2)          nn.Tanh:updateGradInput (a = gradOutput, b = the tanh OUTPUT; LSTM.lua:157)
 *   AOCR_PW_RELU_BWD   y = a * (b > 0)            cudnn.ReLU:updateGradInput where no pool follows (b = the ReLU output)
 *   AOCR_PW_RELU       y = max(a, 0)              cudnn.ReLU:updateOutput on its own (b ignored, may be NULL) */
#define AOCR_PW_ADD 0
#define AOCR_PW_TANH_BWD 1
#define AOCR_PW_RELU_BWD 2
#define AOCR_PW_RELU 3
int aocr_pointwise(void* stream, int32_t op, const float* a_dev, const float* b_dev, float* y_dev, int64_t n);

/* nn.LookupTable (LSTM.lua:55-56): out (n, E) = weight[ids - 1] for n 1-based int32 ids; backward accumulates
 * gradWeight[ids - 1] += gradOutput rows (accGradParameters).  weight / gradWeight (V, E). */
int aocr_lookup_forward(void* stream, const float* weight_dev, const int32_t* ids_dev, float* out_dev, int32_t n, int32_t E);
int aocr_lookup_backward(void* stream, const float* grad_out_dev, const int32_t* ids_dev, float* grad_weight_dev, int32_t n, int32_t E, int32_t V);

/* nn.LogSoftMax + nn.ClassNLLCriterion(weights; PAD weight 0; sizeAverage=false): output_projector.lua:6,
 * criterion.lua:3-8, model.lua:644-648.  logits (rows, ld) ; targets (rows) 1-based.  logp (rows,V) optional,
 * dlogits (rows, ld) optional = grad_scale * w[y] * (softmax - onehot); nll_rows (rows) per-row weighted NLL. */
int aocr_logsoftmax_nll(void* stream, const float* logits_dev, int64_t ld, const int32_t* targets_dev, float* logp_dev,
                        float* dlogits_dev, float* nll_rows_dev, int64_t rows, int32_t V, float grad_scale);

/* topk over beam*V candidates + finished-beam PAD masking, model.lua:399-404,446-458,516.
 * logp (B*kin, V); beam_scores (B,kout) in/out; prev_tok (B*kin) or NULL at t=1 (kin=1).
 * Outputs tokens (B,kout) 1-based, parents (B,kout) 0-based source beam. */
int aocr_beam_select(void* stream, const float* logp_dev, const int32_t* prev_tok_dev, float* beam_scores_dev,
                     int32_t* tokens_dev, int32_t* parents_dev, int32_t B, int32_t kin, int32_t kout, int32_t V);

/* The same selection under the dictionary constraint (model.lua:405-445, 460-513):
 * loc_in_dev (B*kin) trie node of every input beam (ignored at t=1, prev_tok_dev == NULL: all
 * beams start at node 0), loc_out_dev (B,kout) node of every selected beam.  V <= 64. */
int aocr_beam_select_dict(void* stream, const float* logp_dev, const int32_t* prev_tok_dev, float* beam_scores_dev,
                          int32_t* tokens_dev, int32_t* parents_dev, int32_t B, int32_t kin, int32_t kout, int32_t V,
                          const aocr_trie* trie, const int32_t* loc_in_dev, int32_t* loc_out_dev);

/* evalWordErrRate's comparison (utils.lua:136-175) on device: both (B,L) id rows are cut at
 * their first EOS (3) and string.levenshtein (utils.lua:55-94) is taken between them.
 * dist_dev (B) edit distance (0 <=> the word is right, :168-171); target_len_dev (B) or NULL:
 * length of the cut target (the denominator of the edit-distance accuracy, :172, README.md:11). */
int aocr_edit_distance(void* stream, const int32_t* labels_dev, const int32_t* targets_dev, int32_t B, int32_t L,
                       int32_t* dist_dev, int32_t* target_len_dev);

/* ---- data path ----------------------------------------------------------------------------------
 * data_gen.lua:68-79: img = 255 * image.rgb2y(img); img = image.scale(img, imgW, 32) for every decoded image of a batch
 * (all images of a batch share imgW: the loader buckets by width, data_gen.lua:91-99).
 * src_dev: the decoded images back to back, uint8, interleaved HWC with 1 (already gray) or 3 (RGB) channels;
 * desc_dev[i]: where image i starts and its size; out_dev (n_images, 1, out_h, out_w) fp32 in 0..255.
 * Arithmetic: single precision, operation for operation that of torch/image's rgb2y and scaleBilinear (rows to out_w first,
 * then columns to out_h; enlarging interpolates with scale (src-1)/(dst-1), shrinking averages the covered source span). */
typedef struct aocr_image_desc {
  int64_t offset;      /* byte offset of the image inside src_dev */
  int32_t height, width, channels, reserved;
} aocr_image_desc;
int aocr_preprocess_lines(void* stream, const uint8_t* src_dev, const aocr_image_desc* desc_dev, int32_t n_images,
                          int32_t out_h, int32_t out_w, float* out_dev);

#ifdef __cplusplus
}
#endif
#endif /* AOCR_H */
