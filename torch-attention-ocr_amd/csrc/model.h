// model.h -- the aocr_model handle: parameter views, workspace carve-up and the fused step drivers.
#pragma once
#include <string>
#include <vector>
#include "../../include/aocr.h"
#include "ops.h"

namespace aocr {

constexpr int MAXL = 4;          // max LSTM layers per stack
constexpr int LOGIT_LD = 40;     // padded leading dimension of the (rows, vocab) logit buffers (vocab <= 40)

struct ParamEntry { std::string name; int group; int64_t offset; int ndim; int64_t shape[4]; int64_t numel; };
struct Layout { std::vector<ParamEntry> e; int64_t group_off[AOCR_NUM_GROUPS + 1]; };
Layout build_layout(const aocr_config& c);

struct ConvP { float *w, *b, *dw, *db; int cin, cout, ks, pad; };
struct BnP { float *w, *b, *dw, *db, *rm, *rv, *save; int C; };
// bf16 shadows of one recurrent weight matrix W [R][C] (leading dimension ld, first column col0 of the fp32 tensor)
struct ShW { const float* w; int64_t ld; int R, C; bf16_t* wb; bf16_t* wtb; float* wtf; };   // wtf: fp32 transpose [C][R] (fp32 mode)
struct LstmP { float *wi, *bi, *wh, *bh, *dwi, *dbi, *dwh, *dbh; int in; ShW swi, swh; };

struct Arena {
  char* base; size_t off;
  template <class T> T* get(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// data-parallel exchange state (comm.hip): provider 0 = none, 1 = RCCL, 2 = host callback
struct CommState {
  int nranks = 1; bool sync_bn = false; int provider = 0;
  void* rccl = nullptr; void* rccl_bn = nullptr; aocr_allreduce_fn fn = nullptr; void* user = nullptr;
  hipStream_t stream = nullptr; hipEvent_t done = nullptr;
  hipEvent_t wait0 = nullptr, wait1 = nullptr; bool timed = false;   // timing pair around the join of the exchange stream (aocr_comm_exposed_ms)
};

struct Dims {                     // geometry of one step
  int B, H, W, L, T;
  int H1, W1, H2, W2, H4, H6, Ho7, Wo7;
};
bool make_dims(const aocr_config& c, int B, int W, int L, Dims& d);

}  // namespace aocr

struct aocr_model {
  aocr_config cfg;
  hipStream_t s;
  bool bf16;
  float *params, *grads, *bn_state;
  aocr::Layout layout;
  aocr::ConvP conv[8];            // 1..7
  aocr::BnP bn[8];                // 3,5,7
  aocr::LstmP enc[2][aocr::MAXL], dec[aocr::MAXL];
  float *lookup, *dlookup, *wa, *dwa, *wc, *dwc, *wo, *bo, *dwo, *dbo;
  aocr::ShW swa, swc;
  int He, Hd, Le, Ld, E, V;

  // ---- workspace
  size_t ws_bytes;
  // CNN
  float *A1, *A2, *Y3, *A3, *A4, *Y5, *A5, *A6, *Y7, *X;
  uint8_t *idx2, *idx4, *idx6;
  uint16_t* route1 = nullptr; bool route1_valid = false;       // conv1's pooling/ReLU decisions of the last training forward (conv1_forward's route)
  float *G0, *G1, *dX; size_t gmax = 0;   // gmax: floats in G0 / G1
  // bf16 shadows of the contraction operands (bf16 compute mode only; nullptr otherwise)
  aocr::bf16_t *A1b, *A2b, *A3b, *A4b, *A5b, *A6b, *G0b, *G2b = nullptr;
  hipEvent_t cw_map[2] = {nullptr, nullptr}, cw_done[2] = {nullptr, nullptr}, cw_main = nullptr;   // cnn_backward: gradient map ready / filter gradient done per map buffer
  aocr::bf16_t *wb[8], *wtb[8];
  float* wtf[8];                  // fp32 mode: conv taps re-laid as [Cin][tap][Cout] (K-contiguous B operand of the data gradient)
  // bf16 shadows of the recurrent activations / gradients (written by the producing epilogues)
  aocr::bf16_t *Xb, *ehs_b[2][aocr::MAXL], *edz_b[2][aocr::MAXL], *dhs_b[aocr::MAXL], *ddz_b[aocr::MAXL], *out_b, *cat_b, *dpre_b, *dq_b;
  void* bn_scratch; float* bn_save;
  int y16[2] = {0, 0};            // the last forward pass left Y3 / Y5 (pre-BatchNorm maps) as bf16 in the first half of their buffers (conv_forward: y_bf16)
  // encoder [dir][layer]
  float *ezx[2][aocr::MAXL], *ehs[2][aocr::MAXL], *ecs[2][aocr::MAXL], *egates[2][aocr::MAXL], *edz[2][aocr::MAXL], *edc[2][aocr::MAXL];
  float *edxl[2];
  float *context, *dctx;
  aocr::bf16_t* context_b;        // bf16 shadow of the context for the attention kernels (bf16 mode)
  // decoder, teacher-forced (rows = B, time-major)
  // round 4: the embedding side of the first decoder layer through the per-token table (decoder_tf_forward / decoder_backward): sums of d z by token [V][4 Hd], index scratch of the
  // token sort, "the last teacher-forced forward read zx1 from the table", "this API call already has an up-to-date table"
  float* emb_seg = nullptr; int* emb_index = nullptr; bool emb_table = false, tab_valid = false;
  // step_prologue (start of a training step): table / gradient zeroing / weight shadows enqueued on the side stream, and the events the model's stream waits for
  hipEvent_t tab_done = nullptr, zero_done = nullptr, shadow_done = nullptr, shadow2_done = nullptr; bool tab_ready = false, zero_pending = false, shadow_pending = false, shadow2_pending = false;   // shadow2: conv2's taps (the head of the job table), an event of their own
  int shadow_tiles_conv2 = 0;
  bool proj_fused = false, dout_ready = false;      // round 6: this training step's projector runs inside loss_and_dlogits (project_loss: logits + criterion + d out in one launch)
  float* loss_pending = nullptr;      // loss_and_dlogits -> decoder_backward: the loss sum still to be enqueued (behind the BPTT kernel)
  hipEvent_t q_go = nullptr, q_done = nullptr; bool q_pending = false;     // q = W_a h_top of all steps on the side stream (decoder_tf_forward -> decoder_backward)
  hipEvent_t enc_ev = nullptr;   // end of an encoder layer's BPTT: its weight gradients start behind it on the side stream (encoder_backward)
  float *emb_all, *zx1_all, *dhs[aocr::MAXL], *dcs[aocr::MAXL], *dgates[aocr::MAXL], *ddz[aocr::MAXL];
  float *out_all, *cat_all, *q_all, *a_all, *logits, *dlogits, *nll_rows;
  float *dout_proj, *dpre_all, *dcat_all, *ds_all, *dq_all, *demb_all;
  float *dh_rec[aocr::MAXL], *dc_st[aocr::MAXL], *dfeed, *loss_tmp;
  // decode (rows = B*beam)
  float *bc[2][aocr::MAXL], *bh[2][aocr::MAXL], *bfeed[2], *bc_new[aocr::MAXL], *bh_new[aocr::MAXL];
  // round 6: bf16 shadows of the decode launch chain's beam state (bf16 mode), so that its step products take both operands from shadows like the training chain (stepl.h, scores against ctx W_a)
  aocr::bf16_t *bh_b[2][aocr::MAXL] = {}, *bfeed_b[2] = {}, *bh_new_b[aocr::MAXL] = {}, *bcat_b = nullptr;
  float *bzx1, *bzx_tab, *bq, *ba, *bcat, *bout, *blogits, *blogp, *beam_scores;     // bzx_tab [V][4Hd]: per-token first-layer gate input
  int32_t *hist_tok, *hist_par, *tgt_pad, *tge_pad, *trie_loc[2];   // trie_loc: dictionary node of every beam (ping-pong)
  void* sgd_scratch;
  float* wg_part = nullptr; size_t wg_part_floats = 0;     // split-K slabs of the filter gradients (conv_backward_filter)

  std::vector<aocr::ShadowJob> shadow_host;   // bf16 mode: job table of the one-launch weight shadow refresh
  aocr::ShadowJob* shadow_dev; int shadow_tiles;
  hipEvent_t grad_ev[4];          // gradient-ready points of the backward pass (aocr_grad_buckets)
  // stacked encoder (Le >= 2): the layers run as a wavefront of sequence chunks, layer l on lay_s[l] (layer 0 / the top layer of the BPTT on the main stream)
  hipStream_t lay_s[aocr::MAXL] = {}; std::vector<hipEvent_t> lay_ev;
  hipStream_t side2 = nullptr; hipEvent_t side2_done = nullptr;   // second side stream (bias sums, projector gradWeight), joined into `side`
  hipStream_t side = nullptr; hipEvent_t side_go = nullptr, side_done = nullptr; bool side_busy = false;   // side stream of the backward pass (model.hip: decoder_backward)
  int64_t conv5_off;              // offset of cnn.conv5.w in the flat vectors: the CNN group is split there
  aocr::Dims last;                // dims of the last step (for the parity taps)
  int last_valid;
  const float* last_images = nullptr; bool last_train = false;   // the caller's image buffer of the last aocr_train_forward_backward (aocr_profile_kernel replays conv1 on it)
  // cluster encoder kernels (rnn_cluster.hip): exchange buffers, error flag, launch epoch (tags = epoch * 4096 + step)
  unsigned long long *cl_xbuf = nullptr, *cl_pbuf = nullptr, *cl_xtab = nullptr; int* cl_err = nullptr; size_t cl_xbytes = 0, cl_pbytes = 0, cl_tbytes = 0; unsigned cl_epoch = 0;
  unsigned long long *dc_xbuf = nullptr, *dc_xtab = nullptr; size_t dc_xbytes = 0, dc_tbytes = 0; aocr::bf16_t* ctxa_b = nullptr; bool ctxa_fresh = false; /* ctxa_b holds ctx W_a of the CURRENT context (set by the beam pass, consumed by the gold pass of the same decode call) */ bool dgates_il = false; unsigned long long* dc_bxbuf = nullptr; size_t dc_bxbytes = 0; float* dc_pbuf = nullptr;
  // nn.Dropout (LSTM.lua:68-69,116-118), training only: p, threshold ceil(p 2^53), seed, train-step counter; masked copies of the
  // inputs of the layers above the first (decoder per step: dhm, encoder per layer: ehm)
  double drop_p = 0.0; unsigned long long drop_thr = 0, drop_seed = 0, drop_step = 0; bool drop_on = false;
  float* dhm[aocr::MAXL] = {}; aocr::bf16_t* dhm_b[aocr::MAXL] = {}; float* ehm[2][aocr::MAXL] = {}; aocr::bf16_t* ehm_b[2][aocr::MAXL] = {};   // decoder cluster kernel (dec_cluster.hip)
  aocr::CommState comm;
  float* bn_snap = nullptr;        // the BatchNorm running statistics as they were at the start of the current training step (step_prologue); restored on the device by the optimizer call that skips a timed-out step's update
  // per-family HIP-event profile (aocr_profile_enable): a mark = "family `tag` runs from here to the next mark"
  bool prof_on = false;
  std::vector<hipEvent_t> prof_ev; std::vector<int> prof_tag; size_t prof_n = 0;
};

namespace aocr {
const char* comm_unique_id(char id[128]);
const char* comm_init_rccl(aocr_model* m, const char id[128], int nranks, int rank, int sync_bn);
const char* comm_init_callback(aocr_model* m, aocr_allreduce_fn fn, void* user, int nranks, int sync_bn);
void comm_destroy(aocr_model* m);
int comm_allreduce(aocr_model* m, void* buf, int64_t count, int dtype, hipStream_t stream, int channel);
int comm_allreduce_grads(aocr_model* m, float* loss_dev);
inline bool sync_bn_on(const aocr_model* m) { return m->comm.provider != 0 && m->comm.sync_bn; }
// Exchange vs whole-sequence kernels.  Default with a communicator attached: bucket 0 of the gradient exchange is
// held behind the encoder BPTT, so no collective kernel is ever co-resident with a cluster kernel.  AOCR_COMM_EARLY_BUCKET0=1 releases it
// as soon as the decoder's gradients are complete; the encoder cluster launches then keep comm_reserved_cus() compute units free for the
// collective's workgroups (AOCR_COMM_RESERVE_CUS, default 32 = RCCL's channel count on an 8-GPU xGMI node).
inline bool env_on(const char* name) { const char* e = getenv(name); return e && e[0] == '1'; }      // round-4 switches test the VALUE ("0" = off)
inline bool comm_early_bucket0() { return env_on("AOCR_COMM_EARLY_BUCKET0"); }      // read per call, like every dispatch switch
inline bool comm_holds_bucket0(const aocr_model* m) { return m->comm.provider != 0 && !comm_early_bucket0(); }
inline int comm_reserved_cus(const aocr_model* m) {
  if (m->comm.provider == 0 || !comm_early_bucket0()) return 0;
  const char* e = getenv("AOCR_COMM_RESERVE_CUS");
  const int n = e ? atoi(e) : 32;
  return n < 0 ? 0 : (n > 128 ? 128 : n);
}
void prof_mark_slow(aocr_model* m, int tag);
inline void prof_mark(aocr_model* m, int tag) { if (m->prof_on) prof_mark_slow(m, tag); }
void step_prologue(aocr_model* m, size_t grad_bytes);              // start of a training step (capi.hip): gradient zeroing, weight shadows, token table
void build_shadow_jobs(aocr_model* m);                              // after bind_params + model_carve
int model_carve(aocr_model* m, void* base, size_t bytes);          // returns 0 / -1 (too small); base==nullptr: size only
void cnn_forward(aocr_model* m, const float* images, const Dims& d, int training, int update_running);
void encoder_forward(aocr_model* m, const Dims& d);
void decoder_tf_forward(aocr_model* m, const Dims& d, const int32_t* tgt, int64_t st, int64_t sb, bool keep_gates);
void loss_and_dlogits(aocr_model* m, const Dims& d, const int32_t* tge, int64_t st, int64_t sb, float grad_scale, bool want_grad,
                      float* loss_dev);
void backward_all(aocr_model* m, const float* images, const int32_t* tgt, const Dims& d);
void decode_beam(aocr_model* m, const Dims& d, const int32_t* tgt, int beam, int32_t* labels, float* scores,
                 const aocr_trie* trie = nullptr);
}  // namespace aocr
