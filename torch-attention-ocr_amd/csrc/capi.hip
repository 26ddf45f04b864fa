// capi.hip -- the extern "C" surface declared in include/aocr.h.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include "model.h"

using namespace aocr;

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
  char buf[512]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf; return 1;
}
static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
  return 0;
}
#define REQUIRE(cond, ...) do { if (!(cond)) return fail(__VA_ARGS__); } while (0)

static int check_cfg(const aocr_config* c) {
  REQUIRE(c, "config is NULL");
  REQUIRE(c->batch_size >= 1 && c->img_h >= 32 && c->img_h % 16 == 0, "bad batch_size/img_h");
  REQUIRE(c->max_img_w >= 8, "max_img_w too small");
  REQUIRE(c->enc_hidden >= 16 && c->enc_hidden % 16 == 0, "enc_hidden must be a multiple of 16");
  REQUIRE(c->enc_layers >= 1 && c->enc_layers <= MAXL && c->dec_layers >= 1 && c->dec_layers <= MAXL, "layers must be 1..%d", MAXL);
  REQUIRE(c->vocab >= 4 && c->vocab <= LOGIT_LD, "vocab must be 4..%d", LOGIT_LD);
  REQUIRE(c->emb >= 1 && c->max_decoder_l >= 1 && c->max_beam >= 1, "bad emb/max_decoder_l/max_beam");
  REQUIRE(c->compute == AOCR_COMPUTE_F32 || c->compute == AOCR_COMPUTE_BF16, "bad compute type");
  return 0;
}

extern "C" {

const char* aocr_last_error(void) { return g_err.c_str(); }
int aocr_version(void) { return AOCR_VERSION; }

int aocr_param_counts(const aocr_config* cfg, int64_t counts[AOCR_NUM_GROUPS]) {
  if (check_cfg(cfg)) return 1;
  Layout L = build_layout(*cfg);
  for (int g = 0; g < AOCR_NUM_GROUPS; ++g) counts[g] = L.group_off[g + 1] - L.group_off[g];
  return 0;
}
int aocr_param_entry(const aocr_config* cfg, int32_t index, char name[64], int32_t* group, int64_t* offset, int32_t* ndim,
                     int64_t shape[4]) {
  if (check_cfg(cfg)) return -1;
  Layout L = build_layout(*cfg);
  if (index < 0 || index >= (int)L.e.size()) return 1;
  const ParamEntry& e = L.e[index];
  snprintf(name, 64, "%s", e.name.c_str()); *group = e.group; *offset = e.offset; *ndim = e.ndim;
  for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
  return 0;
}
int64_t aocr_bn_state_count(void) { return 2 * (256 + 512 + 512); }

static void bind_params(aocr_model* m) {
  auto find = [&](const std::string& n) -> int64_t {
    for (auto& e : m->layout.e) if (e.name == n) return e.offset;
    return -1;
  };
  static const int convs[7][4] = {{1, 64, 3, 1}, {64, 128, 3, 1}, {128, 256, 3, 1}, {256, 256, 3, 1},
                                  {256, 512, 3, 1}, {512, 512, 3, 1}, {512, 512, 2, 0}};
  float* bnst = m->bn_state; int bi = 0;
  for (int i = 1; i <= 7; ++i) {
    char nm[64];
    ConvP& c = m->conv[i]; c.cin = convs[i - 1][0]; c.cout = convs[i - 1][1]; c.ks = convs[i - 1][2]; c.pad = convs[i - 1][3];
    snprintf(nm, 64, "cnn.conv%d.w", i); int64_t o = find(nm); c.w = m->params + o; c.dw = m->grads + o;
    snprintf(nm, 64, "cnn.conv%d.b", i); o = find(nm); c.b = m->params + o; c.db = m->grads + o;
    if (i == 3 || i == 5 || i == 7) {
      BnP& b = m->bn[i]; b.C = c.cout;
      snprintf(nm, 64, "cnn.bn%d.w", i); o = find(nm); b.w = m->params + o; b.dw = m->grads + o;
      snprintf(nm, 64, "cnn.bn%d.b", i); o = find(nm); b.b = m->params + o; b.db = m->grads + o;
      b.rm = bnst; b.rv = bnst + b.C; bnst += 2 * b.C;
      b.save = m->bn_save + (size_t)bi * 1024; ++bi;
    }
  }
  auto lstm = [&](const char* prefix, LstmP* arr, int layers, int in0, int H) {
    for (int l = 1; l <= layers; ++l) {
      char nm[64]; LstmP& p = arr[l - 1]; p.in = l == 1 ? in0 : H; int64_t o;
      snprintf(nm, 64, "%s.l%d.i2h.w", prefix, l); o = find(nm); p.wi = m->params + o; p.dwi = m->grads + o;
      snprintf(nm, 64, "%s.l%d.i2h.b", prefix, l); o = find(nm); p.bi = m->params + o; p.dbi = m->grads + o;
      snprintf(nm, 64, "%s.l%d.h2h.w", prefix, l); o = find(nm); p.wh = m->params + o; p.dwh = m->grads + o;
      snprintf(nm, 64, "%s.l%d.h2h.b", prefix, l); o = find(nm); p.bh = m->params + o; p.dbh = m->grads + o;
      // recurrent-step views: the part of W_i2h that multiplies a hidden-size input (layer 1 of the decoder: the
      // input-feed columns [E, E+Hd)), and W_h2h
      const int skip = p.in - H;
      p.swi.w = p.wi + (skip > 0 ? skip : 0); p.swi.ld = p.in; p.swi.R = 4 * H; p.swi.C = H;
      if (in0 == 512 && l == 1) { p.swi.w = p.wi; p.swi.C = 512; }                  // encoder layer 1: the whole W_i2h (hoisted GEMMs)
      p.swh.w = p.wh; p.swh.ld = H; p.swh.R = 4 * H; p.swh.C = H;
    }
  };
  lstm("enc_fw", m->enc[0], m->Le, 512, m->He);
  lstm("enc_bw", m->enc[1], m->Le, 512, m->He);
  lstm("dec", m->dec, m->Ld, m->E + (m->cfg.input_feed ? m->Hd : 0), m->Hd);
  int64_t o = find("dec.lookup"); m->lookup = m->params + o; m->dlookup = m->grads + o;
  o = find("dec.attn.wa"); m->wa = m->params + o; m->dwa = m->grads + o;
  o = find("dec.attn.wc"); m->wc = m->params + o; m->dwc = m->grads + o;
  m->swa.w = m->wa; m->swa.ld = m->Hd; m->swa.R = m->Hd; m->swa.C = m->Hd;
  m->swc.w = m->wc; m->swc.ld = 2 * m->Hd; m->swc.R = m->Hd; m->swc.C = 2 * m->Hd;
  o = find("proj.w"); m->wo = m->params + o; m->dwo = m->grads + o;
  o = find("proj.b"); m->bo = m->params + o; m->dbo = m->grads + o;
}

static void init_model_fields(aocr_model* m, const aocr_config* cfg) {
  m->cfg = *cfg; m->bf16 = cfg->compute == AOCR_COMPUTE_BF16;
  m->He = cfg->enc_hidden; m->Hd = 2 * cfg->enc_hidden; m->Le = cfg->enc_layers; m->Ld = cfg->dec_layers;
  m->E = cfg->emb; m->V = cfg->vocab; m->last_valid = 0;
}

size_t aocr_workspace_bytes(const aocr_config* cfg) {
  if (check_cfg(cfg)) return 0;
  aocr_model* m = new (std::nothrow) aocr_model();
  if (!m) return 0;
  init_model_fields(m, cfg);
  size_t r = model_carve(m, nullptr, 0) == 0 ? m->ws_bytes : 0;
  delete m;
  return r;
}

int aocr_model_create(const aocr_config* cfg, float* params_dev, float* grads_dev, float* bn_state_dev, void* workspace_dev,
                      size_t workspace_bytes, void* stream, aocr_model** out) {
  if (check_cfg(cfg)) return 1;
  REQUIRE(params_dev && grads_dev && bn_state_dev && workspace_dev && out, "NULL pointer argument");
  REQUIRE(((uintptr_t)params_dev & 15) == 0 && ((uintptr_t)grads_dev & 15) == 0 && ((uintptr_t)workspace_dev & 255) == 0,
          "params/grads must be 16-byte and the workspace 256-byte aligned");
  aocr_model* m = new (std::nothrow) aocr_model();
  REQUIRE(m, "out of host memory");
  init_model_fields(m, cfg);
  m->s = (hipStream_t)stream; m->params = params_dev; m->grads = grads_dev; m->bn_state = bn_state_dev;
  m->layout = build_layout(*cfg);
  if (model_carve(m, workspace_dev, workspace_bytes) != 0) { delete m; return fail("workspace too small: need %zu bytes", aocr_workspace_bytes(cfg)); }
  bind_params(m);
  build_shadow_jobs(m);
  if (!m->shadow_host.empty() &&
      hipMemcpy(m->shadow_dev, m->shadow_host.data(), m->shadow_host.size() * sizeof(ShadowJob), hipMemcpyHostToDevice) != hipSuccess) {
    delete m; return fail("upload of the shadow job table failed");
  }
  if (m->cl_xbuf) {                                     // tags start at epoch 1: the exchange buffers must not hold a stale match
    hipMemsetAsync(m->cl_xbuf, 0, m->cl_xbytes, m->s); hipMemsetAsync(m->cl_pbuf, 0, m->cl_pbytes, m->s); hipMemsetAsync(m->cl_err, 0, 64, m->s);
    hipMemsetAsync(m->cl_xtab, 0, m->cl_tbytes, m->s);
    if (m->dc_xbuf) { hipMemsetAsync(m->dc_xbuf, 0, m->dc_xbytes, m->s); hipMemsetAsync(m->dc_xtab, 0, m->dc_tbytes, m->s); hipMemsetAsync(m->dc_bxbuf, 0, m->dc_bxbytes, m->s); }
  }
  for (int i = 0; i < 4; ++i)
  {
    const hipError_t e = hipEventCreateWithFlags(&m->grad_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) { aocr_model_destroy(m); return fail("hipEventCreateWithFlags failed: %s", hipGetErrorString(e)); }
  }
  *out = m;
  return 0;
}
int aocr_model_destroy(aocr_model* m) {
  if (m) for (int i = 0; i < 4; ++i) if (m->grad_ev[i]) hipEventDestroy(m->grad_ev[i]);
  if (m && m->side) { hipStreamSynchronize(m->side); hipStreamDestroy(m->side); }
  if (m && m->side_go) hipEventDestroy(m->side_go);
  if (m && m->side_done) hipEventDestroy(m->side_done);
  if (m && m->side2_done) hipEventDestroy(m->side2_done);
  if (m && m->side2) hipStreamDestroy(m->side2);
  if (m) for (hipEvent_t e : {m->cw_map[0], m->cw_map[1], m->cw_done[0], m->cw_done[1], m->cw_main}) if (e) hipEventDestroy(e);
  if (m && m->tab_done) hipEventDestroy(m->tab_done);
  if (m && m->zero_done) hipEventDestroy(m->zero_done);
  if (m && m->shadow_done) hipEventDestroy(m->shadow_done);
  if (m && m->shadow2_done) hipEventDestroy(m->shadow2_done);
  if (m && m->q_go) hipEventDestroy(m->q_go);
  if (m && m->q_done) hipEventDestroy(m->q_done);
  if (m && m->enc_ev) hipEventDestroy(m->enc_ev);
  if (m) for (hipEvent_t e : m->prof_ev) hipEventDestroy(e);
  if (m) for (hipStream_t ls : m->lay_s) if (ls) { hipStreamSynchronize(ls); hipStreamDestroy(ls); }
  if (m) for (hipEvent_t e : m->lay_ev) hipEventDestroy(e);
  if (m) comm_destroy(m);
  delete m; return 0;
}
int aocr_comm_unique_id(char id[128]) {
  REQUIRE(id, "NULL argument");
  if (const char* e = comm_unique_id(id)) return fail("%s", e);
  return 0;
}
int aocr_comm_init_rank(aocr_model* m, const char id[128], int32_t nranks, int32_t rank, int32_t sync_bn) {
  REQUIRE(m && id && nranks >= 1 && rank >= 0 && rank < nranks, "bad arguments");
  REQUIRE(m->comm.provider == 0, "a communicator is already attached (aocr_comm_destroy first)");
  if (const char* e = comm_init_rccl(m, id, nranks, rank, sync_bn)) return fail("%s", e);
  return 0;
}
int aocr_comm_set_callback(aocr_model* m, aocr_allreduce_fn fn, void* user, int32_t nranks, int32_t sync_bn) {
  REQUIRE(m && fn && nranks >= 1, "bad arguments");
  REQUIRE(m->comm.provider == 0, "a communicator is already attached (aocr_comm_destroy first)");
  if (const char* e = comm_init_callback(m, fn, user, nranks, sync_bn)) return fail("%s", e);
  return 0;
}
int aocr_allreduce_grads(aocr_model* m, float* loss_dev) {
  REQUIRE(m, "NULL model");
  REQUIRE(m->comm.provider != 0, "no communicator attached (aocr_comm_init_rank / aocr_comm_set_callback)");
  REQUIRE(m->last_valid, "no train step has been enqueued yet");
  const int rc = comm_allreduce_grads(m, loss_dev);
  REQUIRE(rc == 0, rc == 2 ? "the all-reduce provider reported an error" : "stream / event error in the gradient exchange");
  return 0;
}
int aocr_comm_exposed_ms(aocr_model* m, float* ms) {
  REQUIRE(m && ms, "NULL argument");
  *ms = 0.f;
  if (!m->comm.timed) return 0;
  if (hipEventSynchronize(m->comm.wait1) != hipSuccess || hipEventElapsedTime(ms, m->comm.wait0, m->comm.wait1) != hipSuccess)
    return fail("aocr_comm_exposed_ms: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}
int aocr_comm_destroy(aocr_model* m) { REQUIRE(m, "NULL model"); comm_destroy(m); return 0; }
int aocr_comm_info(aocr_model* m, int32_t* nranks, int32_t* sync_bn, int32_t* provider) {
  REQUIRE(m, "NULL model");
  if (nranks) *nranks = m->comm.provider ? m->comm.nranks : 1;
  if (sync_bn) *sync_bn = sync_bn_on(m) ? 1 : 0;
  if (provider) *provider = m->comm.provider;
  return 0;
}
int aocr_profile_enable(aocr_model* m, int32_t on) { REQUIRE(m, "NULL model"); m->prof_on = on != 0; m->prof_n = 0; return 0; }
int aocr_profile_read(aocr_model* m, float ms[AOCR_PROF_FAMILIES], int32_t* marks) {
  REQUIRE(m && ms, "NULL argument");
  for (int i = 0; i < AOCR_PROF_FAMILIES; ++i) ms[i] = 0.f;
  if (marks) *marks = (int32_t)m->prof_n;
  if (m->prof_n < 2) { m->prof_n = 0; return 0; }
  if (hipEventSynchronize(m->prof_ev[m->prof_n - 1]) != hipSuccess) return fail("hipEventSynchronize failed");
  for (size_t i = 0; i + 1 < m->prof_n; ++i) {
    const int tag = m->prof_tag[i];
    if (tag < 0) continue;                               // a closing mark: the gap to the next entry point is not kernel time
    float t = 0.f; if (hipEventElapsedTime(&t, m->prof_ev[i], m->prof_ev[i + 1]) == hipSuccess) ms[tag] += t;
  }
  m->prof_n = 0;
  return 0;
}
int aocr_model_set_stream(aocr_model* m, void* stream) { REQUIRE(m, "NULL model"); m->s = (hipStream_t)stream; return 0; }

static int step_dims(aocr_model* m, int32_t B, int32_t W, int32_t L, Dims& d) {
  REQUIRE(m, "NULL model");
  REQUIRE(B >= 1 && B <= m->cfg.batch_size, "B=%d outside 1..%d", B, m->cfg.batch_size);
  REQUIRE(W >= 8 && W <= m->cfg.max_img_w, "W=%d outside 8..%d", W, m->cfg.max_img_w);
  REQUIRE(L >= 1 && L <= m->cfg.max_decoder_l, "max_decoder_l (%d) < target_l (%d)!", m->cfg.max_decoder_l, L);   // model.lua:264
  REQUIRE(make_dims(m->cfg, B, W, L, d), "image too small for the CNN");
  return 0;
}

int aocr_cluster_status(aocr_model* m, int32_t* code) {
  REQUIRE(m && code, "NULL argument");
  *code = 0;
  if (!m->cl_err) return 0;
  int32_t w[16] = {0};           // word 0: a code no optimizer call has consumed yet (a decode call's, or feval without an update); CL_ERR_STICKY: the last code an optimizer call skipped its update on
  if (hipMemcpyAsync(w, m->cl_err, sizeof(w), hipMemcpyDeviceToHost, m->s) != hipSuccess || hipStreamSynchronize(m->s) != hipSuccess)
    return fail("aocr_cluster_status: %s", hipGetErrorString(hipGetLastError()));
  *code = w[CL_ERR_STICKY] != 0 ? w[CL_ERR_STICKY] : w[0];
  if (*code != 0) {
    hipMemsetAsync(m->cl_err, 0, sizeof(int32_t), m->s); hipMemsetAsync(m->cl_err + CL_ERR_LATCH, 0, 2 * sizeof(int32_t), m->s);      // read and clear: the next call reports the steps after this one
    // (the skipped step's move of the BatchNorm running statistics was taken back by the optimizer call that skipped the update --
    //  sgd_update_kernel / adadelta_kernel restore the snapshot of step_prologue -- so a repeat of the batch moves them exactly once,
    //  whenever the host polls)
  }
  return 0;
}

int aocr_set_dropout(aocr_model* m, double p, uint64_t seed, uint64_t train_step) {
  REQUIRE(m, "NULL model");
  REQUIRE(p >= 0.0 && p < 1.0, "dropout p=%g outside [0, 1)", p);
  m->drop_p = p; m->drop_seed = seed; m->drop_step = train_step;
  m->drop_thr = p > 0.0 ? (unsigned long long)ceil(p * 9007199254740992.0) : 0ull;       // keep <=> (r >> 11) >= p 2^53
  return 0;
}

int aocr_train_forward_backward(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev,
                                int32_t B, int32_t W, int32_t L, float grad_scale, float* loss_dev) {
  Dims d; if (step_dims(m, B, W, L, d)) return 1;
  REQUIRE(images_dev && targets_dev && targets_eval_dev, "NULL input");
  prof_mark(m, AOCR_PROF_OTHER);
  step_prologue(m, (size_t)m->layout.group_off[AOCR_NUM_GROUPS] * sizeof(float));      // model.lua:637-639 (zeroGradParameters) + the step's parameter-only work
  m->drop_on = m->drop_thr != 0;                                          // nn.Dropout is active in training() mode only (model.lua:284)
  cnn_forward(m, images_dev, d, 1, 1);
  encoder_forward(m, d);
  decoder_tf_forward(m, d, targets_dev, 1, L, true);
  loss_and_dlogits(m, d, targets_eval_dev, 1, L, grad_scale, true, loss_dev);
  backward_all(m, images_dev, targets_dev, d);
  m->drop_on = false;
  prof_mark(m, -1);
  m->last = d; m->last_valid = 1; m->last_images = images_dev; m->last_train = true; m->tab_valid = false;
  return check_launch("aocr_train_forward_backward");
}

static int64_t conv5_offset(const Layout& l) {
  for (const ParamEntry& e : l.e) if (e.name == "cnn.conv5.w") return e.offset;
  return 0;
}
int aocr_grad_buckets(const aocr_config* cfg, int64_t begin[AOCR_GRAD_BUCKETS], int64_t end[AOCR_GRAD_BUCKETS]) {
  if (check_cfg(cfg)) return 1;
  REQUIRE(begin && end, "NULL argument");
  const Layout l = build_layout(*cfg);
  const int64_t c5 = conv5_offset(l);
  begin[0] = l.group_off[3]; end[0] = l.group_off[AOCR_NUM_GROUPS];       // decoder + projector
  begin[1] = l.group_off[1]; end[1] = l.group_off[3];                     // both encoder directions
  begin[2] = c5;             end[2] = l.group_off[1];                     // CNN from conv5 upwards
  begin[3] = 0;              end[3] = c5;                                 // conv1 .. conv4 (+ bn3)
  return 0;
}
int aocr_stream_wait_grads(aocr_model* m, int32_t bucket, void* stream) {
  REQUIRE(m, "NULL model");
  REQUIRE(bucket >= 0 && bucket < AOCR_GRAD_BUCKETS, "bucket %d outside 0..%d", bucket, AOCR_GRAD_BUCKETS - 1);
  REQUIRE(m->last_valid, "no train step has been enqueued yet");
  if (hipStreamWaitEvent((hipStream_t)stream, m->grad_ev[bucket], 0) != hipSuccess) return fail("hipStreamWaitEvent failed");
  return 0;
}

int aocr_sgd_step(aocr_model* m, float lr, float clip, float* norms_dev) {
  REQUIRE(m, "NULL model");
  prof_mark(m, AOCR_PROF_SGD);
  sgd_clip_update(m->s, m->params, m->grads, m->layout.group_off, lr, clip, norms_dev, m->sgd_scratch, m->cl_err, m->bn_snap ? m->bn_state : nullptr, m->bn_snap, (int)aocr_bn_state_count());
  prof_mark(m, -1);
  return check_launch("aocr_sgd_step");
}

int aocr_adadelta_step(aocr_model* m, float rho, float eps, float weight_decay, float* state_dev) {
  REQUIRE(m && state_dev, "NULL argument");
  REQUIRE(rho >= 0.f && rho < 1.f && eps > 0.f, "rho=%g eps=%g out of range", rho, eps);
  const int64_t n = m->layout.group_off[AOCR_NUM_GROUPS];
  adadelta_update(m->s, m->params, m->grads, state_dev, state_dev + n, n, rho, eps, weight_decay, m->cl_err, m->bn_snap ? m->bn_state : nullptr, m->bn_snap, (int)aocr_bn_state_count());
  return check_launch("aocr_adadelta_step");
}

int aocr_forward_logits(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev,
                        int32_t B, int32_t W, int32_t L, int32_t training, float* logits_dev, float* loss_dev) {
  Dims d; if (step_dims(m, B, W, L, d)) return 1;
  REQUIRE(images_dev && targets_dev, "NULL input");
  m->tab_valid = false;                                                   // (the per-token table of an earlier call may be stale: the parameters may have moved)
  cnn_forward(m, images_dev, d, training, 0);
  encoder_forward(m, d);
  decoder_tf_forward(m, d, targets_dev, 1, L, false);
  if (targets_eval_dev) loss_and_dlogits(m, d, targets_eval_dev, 1, L, 0.f, false, loss_dev);
  if (logits_dev) copy2d(m->s, m->logits, LOGIT_LD, logits_dev, m->V, L * B, m->V);
  m->last = d; m->last_valid = 1; m->last_train = false;
  return check_launch("aocr_forward_logits");
}

__global__ void pad_targets_kernel(const int32_t* src, int32_t* dst, int B, int L, int Lt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Lt) return;
  int b = i / Lt, t = i - b * Lt;
  dst[i] = t < L ? src[b * L + t] : 1;                                    // model.lua:267-272: pad with PAD
}

int aocr_decode(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B,
                int32_t W, int32_t L, int32_t beam, int32_t* labels_dev, float* scores_dev, float* gold_scores_dev, float* loss_dev) {
  return aocr_decode_dict(m, images_dev, targets_dev, targets_eval_dev, B, W, L, beam, nullptr, labels_dev, scores_dev, gold_scores_dev,
                          loss_dev);
}
static int check_trie(const aocr_trie* t, int V) {
  REQUIRE(t->child_mask_dev && t->child_base_dev && (t->child_dev || t->n_edges == 0), "trie: NULL array");
  REQUIRE(t->n_nodes >= 1 && t->n_edges >= 0, "trie: n_nodes=%d n_edges=%d", t->n_nodes, t->n_edges);
  REQUIRE(V <= 64, "dictionary decoding needs target_vocab_size <= 64 (got %d)", V);
  return 0;
}
int aocr_decode_dict(aocr_model* m, const float* images_dev, const int32_t* targets_dev, const int32_t* targets_eval_dev, int32_t B,
                     int32_t W, int32_t L, int32_t beam, const aocr_trie* trie, int32_t* labels_dev, float* scores_dev,
                     float* gold_scores_dev, float* loss_dev) {
  Dims d; if (step_dims(m, B, W, L, d)) return 1;
  REQUIRE(images_dev && targets_dev && targets_eval_dev && labels_dev && scores_dev, "NULL argument");
  if (trie && check_trie(trie, m->V)) return 1;
  if (beam > m->V) beam = m->V;                                           // model.lua:229
  REQUIRE(beam >= 1 && beam <= m->cfg.max_beam, "beam=%d outside 1..%d", beam, m->cfg.max_beam);
  const int Lt = m->cfg.max_decoder_l;                                    // model.lua:273: always max_decoder_l steps (S8)
  hipLaunchKernelGGL(pad_targets_kernel, dim3(cdiv((int64_t)B * Lt, 256)), dim3(256), 0, m->s, targets_dev, m->tgt_pad, B, L, Lt);
  hipLaunchKernelGGL(pad_targets_kernel, dim3(cdiv((int64_t)B * Lt, 256)), dim3(256), 0, m->s, targets_eval_dev, m->tge_pad, B, L, Lt);
  d.L = Lt;
  prof_mark(m, AOCR_PROF_OTHER);
  m->tab_valid = false;                                                   // (a decode call keeps its weight shadows / token table in line: on the side stream they measured 2 % slower, 1.73 -> 1.76 ms per call)
  cnn_forward(m, images_dev, d, 0, 0);                                    // model.lua:280-281: evaluate()
  encoder_forward(m, d);
  prof_mark(m, AOCR_PROF_DECODE);
  decode_beam(m, d, m->tgt_pad, beam, labels_dev, scores_dev, trie);
  // gold pass, model.lua:589-627.  The reference runs it over all max_decoder_l steps of the PAD-filled target (model.lua:266-274); a step
  // whose targets are all PAD adds nothing to the loss (criterion weight 0, criterion.lua:4-5) nor to the gold scores (model.lua:614-618)
  // and no later code reads the decoder state, so the L steps the caller's targets span give identical outputs.
  d.L = L < Lt ? L : Lt;
  decoder_tf_forward(m, d, m->tgt_pad, 1, Lt, false);
  prof_mark(m, AOCR_PROF_OTHER);
  loss_and_dlogits(m, d, m->tge_pad, 1, Lt, 0.f, false, loss_dev);
  if (gold_scores_dev) gold_scores(m->s, m->nll_rows, gold_scores_dev, d.L, B);
  prof_mark(m, -1);
  m->last = d; m->last_valid = 1; m->last_train = false; m->tab_valid = false;
  return check_launch(trie ? "aocr_decode_dict" : "aocr_decode");
}

int aocr_get_tensor(aocr_model* m, const char* name, const void** ptr_dev, int32_t* ndim, int64_t shape[4]) {
  REQUIRE(m && name && ptr_dev && ndim && shape, "NULL argument");
  REQUIRE(m->last_valid, "no step has run yet");
  const Dims& d = m->last; std::string n = name;
  for (int i = 0; i < 4; ++i) shape[i] = 1;
  if (n == "feats") { *ptr_dev = m->X; *ndim = 3; shape[0] = d.T; shape[1] = d.B; shape[2] = 512; }
  else if (n == "dfeats") { *ptr_dev = m->dX; *ndim = 3; shape[0] = d.T; shape[1] = d.B; shape[2] = 512; }
  else if (n == "context") { *ptr_dev = m->context; *ndim = 3; shape[0] = d.B; shape[1] = d.T; shape[2] = m->Hd; }
  else if (n == "dcontext") { *ptr_dev = m->dctx; *ndim = 3; shape[0] = d.B; shape[1] = d.T; shape[2] = m->Hd; }
  else if (n == "logits") { *ptr_dev = m->logits; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = LOGIT_LD; }
  else if (n == "dlogits") { *ptr_dev = m->dlogits; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = LOGIT_LD; }      // (after a training step)
  else if (n == "dout_proj") { *ptr_dev = m->dout_proj; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = m->Hd; }
  else if (n == "outs") { *ptr_dev = m->out_all + (size_t)d.B * m->Hd; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = m->Hd; }
  else if (n == "conv1") { if (m->bf16) bf16_to_f32(m->s, m->A1b, m->A1, (int64_t)d.B * d.H1 * d.W1 * 64); *ptr_dev = m->A1; *ndim = 4; shape[0] = d.B; shape[1] = d.H1; shape[2] = d.W1; shape[3] = 64; }
  else if (n == "conv2") { if (m->bf16) bf16_to_f32(m->s, m->A2b, m->A2, (int64_t)d.B * d.H2 * d.W2 * 128); *ptr_dev = m->A2; *ndim = 4; shape[0] = d.B; shape[1] = d.H2; shape[2] = d.W2; shape[3] = 128; }
  else if (n == "conv6") { if (m->bf16) bf16_to_f32(m->s, m->A6b, m->A6, (int64_t)d.B * d.H6 * d.W2 * 512); *ptr_dev = m->A6; *ndim = 4; shape[0] = d.B; shape[1] = d.H6; shape[2] = d.W2; shape[3] = 512; }
  // parity aids: the ReLU / max-pool DECISIONS of the forward pass (tests impose them on the fp64 oracle: tests/test_step_gpu.py)
  else if (n == "conv3") { if (m->bf16) bf16_to_f32(m->s, m->A3b, m->A3, (int64_t)d.B * d.H2 * d.W2 * 256); *ptr_dev = m->A3; *ndim = 4; shape[0] = d.B; shape[1] = d.H2; shape[2] = d.W2; shape[3] = 256; }
  else if (n == "conv4") { if (m->bf16) bf16_to_f32(m->s, m->A4b, m->A4, (int64_t)d.B * d.H4 * d.W2 * 256); *ptr_dev = m->A4; *ndim = 4; shape[0] = d.B; shape[1] = d.H4; shape[2] = d.W2; shape[3] = 256; }
  else if (n == "conv5") { if (m->bf16) bf16_to_f32(m->s, m->A5b, m->A5, (int64_t)d.B * d.H4 * d.W2 * 512); *ptr_dev = m->A5; *ndim = 4; shape[0] = d.B; shape[1] = d.H4; shape[2] = d.W2; shape[3] = 512; }
  else if (n == "idx2") { u8_to_f32(m->s, m->idx2, m->G0, (int64_t)d.B * d.H2 * d.W2 * 128); *ptr_dev = m->G0; *ndim = 4; shape[0] = d.B; shape[1] = d.H2; shape[2] = d.W2; shape[3] = 128; }
  else if (n == "idx4") { u8_to_f32(m->s, m->idx4, m->G0, (int64_t)d.B * d.H4 * d.W2 * 256); *ptr_dev = m->G0; *ndim = 4; shape[0] = d.B; shape[1] = d.H4; shape[2] = d.W2; shape[3] = 256; }
  else if (n == "idx6") { u8_to_f32(m->s, m->idx6, m->G0, (int64_t)d.B * d.H6 * d.W2 * 512); *ptr_dev = m->G0; *ndim = 4; shape[0] = d.B; shape[1] = d.H6; shape[2] = d.W2; shape[3] = 512; }
  else if (n == "enc_dz0" || n == "enc_dz1") {           // debugging aid: bf16 d z of the top encoder layer, direction 0 / 1
    const int dir = n == "enc_dz1"; const int64_t cnt = (int64_t)d.T * d.B * 4 * m->He;
    REQUIRE(m->edz_b[dir][m->Le - 1], "no bf16 d z in this mode");
    bf16_to_f32(m->s, m->edz_b[dir][m->Le - 1], m->G0, cnt);
    *ptr_dev = m->G0; *ndim = 3; shape[0] = d.T; shape[1] = d.B; shape[2] = 4 * m->He;
  }
  else if (n == "enc_cs0") { *ptr_dev = m->ecs[0][m->Le - 1]; *ndim = 3; shape[0] = d.T + 2; shape[1] = d.B; shape[2] = m->He; }
  else if (n == "enc_gates0") { *ptr_dev = m->egates[0][m->Le - 1]; *ndim = 3; shape[0] = d.T; shape[1] = d.B; shape[2] = 4 * m->He; }
  else if (n == "cl_err") { *ptr_dev = m->cl_err; *ndim = 1; shape[0] = 8; REQUIRE(m->cl_err, "no cluster kernels in this configuration"); }
  else if (n == "dc_stamps") { REQUIRE(m->dc_xtab, "no decoder cluster kernel in this configuration"); *ptr_dev = m->dc_xtab + (size_t)((d.B + 31) / 32) * 32; *ndim = 1; shape[0] = 32; }   // 16 x u64 cycle counters (AOCR_DC_STAMPS=1) viewed as 16 floats
  else if (n == "ds_all") { *ptr_dev = m->ds_all; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = d.T; }          // debugging aids: decoder BPTT intermediates
  else if (n == "dq_all") { *ptr_dev = m->dq_all; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = m->Hd; }
  else if (n == "dcat_all") { *ptr_dev = m->dcat_all; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = 2 * m->Hd; }
  else if (n == "dh_rec0" || n == "dh_rec1" || n == "dc_st0" || n == "dc_st1" || n == "dfeed0") {      // debugging aids: gradients of the decoder's initial state
    const int l = n.back() - '0'; *ptr_dev = n[1] == 'f' ? m->dfeed : (n[1] == 'h' ? m->dh_rec[l] : m->dc_st[l]); *ndim = 2; shape[0] = d.B; shape[1] = m->Hd; }
  else if (n == "dpre_all") { *ptr_dev = m->dpre_all; *ndim = 3; shape[0] = d.L; shape[1] = d.B; shape[2] = m->Hd; }
  else if (n == "dc_bstamps") { REQUIRE(m->dc_xtab, "no decoder cluster kernel in this configuration"); *ptr_dev = m->dc_xtab + (size_t)((d.B + 31) / 32) * 32 + 16; *ndim = 1; shape[0] = 32; }
  else if (n == "dc_times") { REQUIRE(m->cl_err, "no cluster kernels in this configuration"); *ptr_dev = m->cl_err + 16 + 2048; *ndim = 1; shape[0] = 32 * 4 * 16 * 2; }   // 32 members x 4 gathers x 16 x u64 (10 ns ticks)
  else if (n == "g0") {                                   // debugging aid (AOCR_DBG_STOP=1|2): the gradient map the CNN backward pass stopped at
    const char* e = getenv("AOCR_DBG_STOP"); const int stop = e ? atoi(e) : 0;
    const int64_t cnt = stop == 1 ? (int64_t)d.B * d.T * 512 : (int64_t)d.B * d.H4 * d.W2 * 512;
    if (m->bf16) bf16_to_f32(m->s, m->G0b, m->G0, cnt);
    *ptr_dev = m->G0; *ndim = 4; shape[0] = d.B; shape[1] = stop == 1 ? d.Ho7 : d.H4; shape[2] = stop == 1 ? d.Wo7 : d.W2; shape[3] = 512;
  }
  else return fail("unknown tensor '%s'", name);
  return 0;
}

int aocr_profile_kernel(aocr_model* m, int32_t which, int32_t iters, float* ms_per_launch, double* flops_per_launch) {
  REQUIRE(m && ms_per_launch && flops_per_launch, "NULL argument");
  REQUIRE(m->last_valid, "run a step first");
  REQUIRE(which >= 0 && which <= AOCR_PK_LAST, "unknown kernel id %d", which);
  REQUIRE(iters >= 1, "iters must be >= 1");
  REQUIRE(which < 2 || (m->bf16 && m->last_images && m->last_train), "the HBM-bound kernel ids replay the bf16 TRAINING step: run aocr_train_forward_backward in bf16 mode first");
  const Dims& d = m->last;
  const int64_t n5 = (int64_t)d.B * d.H4 * d.W2;            // pixels of the conv5 / conv6 maps (512 channels)
  // the replays write where the step writes: guard the buffers they assume
  REQUIRE((which != 1 && which != AOCR_PK_SPLITK) || (size_t)512 * 4608 <= m->gmax, "kernel id %d sums conv6's filter gradient (512 x 4608 floats) into the gradient-map scratch, which holds %zu floats in this configuration", which, m->gmax);
  REQUIRE((which != AOCR_PK_BN_FWD && which != AOCR_PK_BN_BWD && which != AOCR_PK_UNPOOL) || (n5 % 256 == 0 && n5 / 256 <= 512), "kernel id %d replays conv5's BatchNorm with the epilogue's 256-pixel statistics chunks: %lld pixels are not a multiple of 256 / more than 512 chunks", which, (long long)n5);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipStream_t s = m->s; const int B = d.B;
  constexpr size_t SLAB = (size_t)4 << 20;
  double bytes = 0.0;
  // bf16 data-gradient maps (conv_backward_data's dx16, round 4): what the step's BatchNorm backward / un-pool passes read when the switch is on
  const char* const nd = getenv("AOCR_DX16");
  const bool dx16 = (nd && nd[0] == '1') && !getenv("AOCR_BN_PARTIAL_OLD") && !getenv("AOCR_UNPOOL4") && !getenv("AOCR_HALO8") && B * d.H4 * d.W2 % 256 == 0;
  const bf16_t* const g1h = dx16 ? reinterpret_cast<const bf16_t*>(m->G1) : nullptr;
  const double dab = dx16 ? 2.0 : 4.0;                    // bytes per element of d A / d(pooled)
  // ids >= 2: the bandwidth-bound kernels, each replayed on the buffers the last TRAINING step left, with the arguments cnn_forward /
  // cnn_backward / decoder_backward pass (bf16 mode).  Outputs land where the step puts them (activations, gradient maps, gradient
  // vector): the model's taps and gradients are UNDEFINED afterwards until the next step.  flops_per_launch returns ALGORITHMIC BYTES.
  auto run = [&]() {
    switch (which) {
    case 0:
      if (const char* const pl = getenv("AOCR_PROBE_LAYER")) {   // debugging aid (tools/debug/conv_probe.py): the tagged launch of conv3 / conv4 / conv5 forward instead, as cnn_forward makes it
        int bnc = 0, y16 = 0; const int l = atoi(pl);
        if (l == 3) { conv_forward(m->s, m->bf16, m->A2, m->conv[3].w, m->conv[3].b, m->Y3, nullptr, d.B, d.H2, d.W2, 128, 256, 3, 1, 0, 0, m->A2b, m->wb[3], nullptr, 1, nullptr, nullptr, nullptr, (double*)m->bn_scratch, &bnc, &y16); break; }
        if (l == 4) { conv_forward(m->s, m->bf16, m->A3, m->conv[4].w, m->conv[4].b, nullptr, m->idx4, d.B, d.H2, d.W2, 256, 256, 3, 1, 1, 2, m->A3b, m->wb[4], m->A4b, 1); break; }
        if (l == 5) { conv_forward(m->s, m->bf16, m->A4, m->conv[5].w, m->conv[5].b, m->Y5, nullptr, d.B, d.H4, d.W2, 256, 512, 3, 1, 0, 0, m->A4b, m->wb[5], nullptr, 1, nullptr, nullptr, nullptr, (double*)m->bn_scratch, &bnc, &y16); break; }
      }
      conv_forward(m->s, m->bf16, m->A5, m->conv[6].w, m->conv[6].b, m->bf16 ? nullptr : m->A6, m->idx6, d.B, d.H4, d.W2, 512, 512, 3, 1, 1, 2, m->A5b, m->wb[6],
                   m->A6b, 1);                              // exactly the launch cnn_forward makes for conv6 (distinct symbol: TAG = 1)
      break;
    case 1:                                                 // conv6 filter gradient as backward_all launches it (split-K slabs + their sum), summed into scratch
      conv_backward_filter(m->s, m->bf16, m->A5, m->G0, m->G1, nullptr, d.B, d.H4, d.W2, 512, 512, 3, 1, m->A5b, m->G0b, m->wg_part, m->wg_part_floats, 1);
      break;
    case AOCR_PK_CONV1_FWD:                                 // R1 + R2: reads the image (fp32), writes the pooled 64-channel map as bf16
      conv1_forward(s, m->last_images, m->conv[1].w, m->conv[1].b, nullptr, B, d.H, d.W, m->A1b, m->route1_valid ? m->route1 : nullptr);
      bytes = (double)B * d.H * d.W * 4 + (double)B * d.H1 * d.W1 * 64 * 2 + (m->route1_valid ? (double)conv1_route_elems(B, d.H, d.W) * 2 : 0.0);
      break;
    case AOCR_PK_CONV1_BWD:                                 // reads the image and d(pooled map) (fp32, conv2's data gradient); writes 640 numbers
      conv1_backward(s, m->last_images, m->conv[1].w, m->conv[1].b, m->G1, m->conv[1].dw, m->conv[1].db, B, d.H, d.W,
                     (size_t)B * d.H1 * d.W1 * 128 >= (size_t)4096 * 640 ? m->G0 + 6 * SLAB : nullptr, nullptr, m->route1_valid ? m->route1 : nullptr);
      bytes = (double)B * d.H * d.W * 4 + (double)B * d.H1 * d.W1 * 64 * 4 + (m->route1_valid ? (double)conv1_route_elems(B, d.H, d.W) * 2 : 0.0);
      break;
    case AOCR_PK_BN_FWD:                                    // conv5's BatchNorm + ReLU as the step runs it: statistics from the conv epilogue -> finalize + ONE pass: fp32 y in, bf16 out
      bn_relu_forward(s, m->Y5, nullptr, m->bn[5].w, m->bn[5].b, m->bn[5].rm, m->bn[5].rv, m->bn[5].save, m->bn_scratch, n5, 512, 1, 0, 0, m->A5b, nullptr, (int)(n5 / 256));
      bytes = (double)n5 * 512 * (4 + 2);
      break;
    case AOCR_PK_BN_BWD:                                    // conv5's BatchNorm backward: sums pass (x fp32, d A fp32, mask bf16) + apply pass (the same + bf16 d x out)
      bn_relu_backward(s, m->Y5, m->A5, m->G1, m->bn[5].w, m->bn[5].save, nullptr, m->bn[5].dw, m->bn[5].db, m->bn_scratch, n5, 512, 0, m->G0b, m->A5b,
                       m->conv[5].db, m->G0 + 2 * SLAB, nullptr, nullptr, nullptr, g1h);
      bytes = (double)n5 * 512 * ((4 + dab + 2) + (4 + dab + 2 + 2));
      break;
    case AOCR_PK_UNPOOL:                                    // (2,1) un-pool + ReLU backward of conv6: d(pooled) fp32 + arg-max (1 B) + pooled mask (bf16) in, bf16 gradient of the un-pooled map out
      unpool_relu_backward(s, m->G1, m->A6, m->idx6, nullptr, B, d.H4, d.W2, 512, 2, m->G0b, m->conv[6].db, m->G0 + 1 * SLAB, m->A6b, nullptr, g1h);
      bytes = (double)B * d.H6 * d.W2 * 512 * (dab + 1 + 2) + (double)n5 * 512 * 2;
      break;
    case AOCR_PK_ATTN_DCTX:                                 // d(context) of all L steps in one pass (model.lua:652-653): a, d s (L,B,T), d c, q (L,B,Hd) in; (B,T,Hd) fp32 out
      attention_dctx(s, m->a_all, m->ds_all, m->dcat_all, 2 * m->Hd, m->q_all, m->dctx, d.L, B, d.T, m->Hd);
      bytes = (double)d.L * B * (2.0 * d.T + 2.0 * m->Hd) * 4 + (double)B * d.T * m->Hd * 4;
      break;
    case AOCR_PK_SPLITK:                                    // the sum of conv6's seven split-K slabs (512 x 4608 fp32 each) into the gradient
      {
        const int ksl = (int)std::min<size_t>(8, m->wg_part_floats / ((size_t)512 * 4608));       // 32 tiles of 256 x 288 on 256 units: 8 k ranges (conv_backward_filter, conv_wgrad_halo_kernel)
        if (ksl >= 1) splitk_reduce(s, m->wg_part, ksl, (size_t)512 * 4608, m->G1);
        bytes = (double)512 * 4608 * 4 * (ksl + 2);
      }
      break;
    }
  };
  run();
  hipEventRecord(e0, m->s);
  for (int i = 0; i < iters; ++i) run();
  hipEventRecord(e1, m->s);
  hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  *ms_per_launch = ms / iters;
  if (which == 0 && getenv("AOCR_PROBE")) {
    unsigned long long pr[8] = {0}; kprobe_read(pr);
    if (getenv("AOCR_PROBE_LAYER")) for (int k = 1; k >= 0; --k) if (pr[4 * k + 1]) fprintf(stderr, "[aocr] probe: workgroup %d of the tagged launch: start -> K loop %.1f us, K loop %.1f us (%llu cycles = %.2f GHz), whole workgroup (last store acknowledged) %.1f us\n", k ? 17 : 300, pr[4 * k + 2] / 100.0, pr[4 * k + 1] / 100.0, pr[4 * k], pr[4 * k] / (pr[4 * k + 1] * 10.0) , pr[4 * k + 3] / 100.0);
    for (int k = 0; k < 2; ++k) if (pr[4 * k + 1]) fprintf(stderr, "[aocr] probe (%s halo kernel, conv6 forward): K loop of one workgroup %llu shader cycles in %.1f us = %.2f GHz\n", k ? "4-wave" : "8-wave", pr[4 * k], pr[4 * k + 1] / 100.0, pr[4 * k] / (pr[4 * k + 1] * 10.0)), fprintf(stderr, "[aocr]   prologue %.1f us, whole workgroup %.1f us (stores acknowledged)\n", pr[4 * k + 2] / 100.0, pr[4 * k + 3] / 100.0);
  }
  *flops_per_launch = which < 2 ? 2.0 * (double)d.B * d.H4 * d.W2 * 512.0 * (9.0 * 512.0) : bytes;
  return check_launch("aocr_profile_kernel");
}

// ---------------------------------------------------------------------------------------------
// module-level entry points
// ---------------------------------------------------------------------------------------------
int aocr_gemm(void* stream, int32_t compute, const float* A_dev, int64_t lda, int32_t a_kmajor, const float* B_dev, int64_t ldb,
              int32_t b_kmajor, float* C_dev, int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias_dev, int32_t accumulate) {
  REQUIRE(A_dev && B_dev && C_dev && M >= 0 && N >= 0 && K >= 0, "bad gemm arguments");
  REQUIRE((accumulate & ~7) == 0, "unknown gemm flag bits %d", accumulate);
  const int fl = ((accumulate & AOCR_GEMM_ACCUMULATE) ? EP_ACCUM : 0) | ((accumulate & AOCR_GEMM_RELU) ? EP_RELU : 0) | ((accumulate & AOCR_GEMM_TANH) ? EP_TANH : 0);
  REQUIRE(!((fl & EP_ACCUM) && (fl & (EP_RELU | EP_TANH))), "an activation on an accumulating product is not a module of the path");
  int rc = gemm((hipStream_t)stream, compute == AOCR_COMPUTE_BF16, A_dev, lda, a_kmajor != 0, B_dev, ldb, b_kmajor != 0, C_dev, ldc,
                M, N, K, bias_dev, nullptr, fl);
  REQUIRE(rc == 0, "gemm: the (A column-major, B row-major-by-K) combination is not supported");
  return check_launch("aocr_gemm");
}
int aocr_pointwise(void* stream, int32_t op, const float* a_dev, const float* b_dev, float* y_dev, int64_t n) {
  REQUIRE(a_dev && y_dev && n >= 0 && op >= 0 && op <= 3 && (b_dev || op == AOCR_PW_RELU), "bad arguments");
  pointwise((hipStream_t)stream, op, a_dev, b_dev, y_dev, n);
  return check_launch("aocr_pointwise");
}
int aocr_lookup_forward(void* stream, const float* weight_dev, const int32_t* ids_dev, float* out_dev, int32_t n, int32_t E) {
  REQUIRE(weight_dev && ids_dev && out_dev && n >= 0 && E > 0, "bad arguments");
  embedding_gather((hipStream_t)stream, weight_dev, ids_dev, 0, 1, out_dev, 1, n, E);
  return check_launch("aocr_lookup_forward");
}
int aocr_lookup_backward(void* stream, const float* grad_out_dev, const int32_t* ids_dev, float* grad_weight_dev, int32_t n, int32_t E, int32_t V) {
  REQUIRE(grad_out_dev && ids_dev && grad_weight_dev && n >= 0 && E > 0 && E <= 256 && V > 0, "bad arguments");
  embedding_scatter_accum((hipStream_t)stream, grad_out_dev, ids_dev, 0, 1, grad_weight_dev, 1, n, E, V);
  return check_launch("aocr_lookup_backward");
}
int aocr_conv2d_forward(void* stream, int32_t compute, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                        uint8_t* idx_dev, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad,
                        int32_t relu, int32_t pool) {
  REQUIRE(x_dev && w_dev && y_dev, "NULL argument");
  REQUIRE(Cin % 16 == 0 && Cout % 16 == 0, "Cin and Cout must be multiples of 16 (the 1-channel first layer has its own entry point)");
  REQUIRE(pool >= 0 && pool <= 2 && (pool == 0 || idx_dev), "bad pool mode / missing idx");
  conv_forward((hipStream_t)stream, compute == AOCR_COMPUTE_BF16, x_dev, w_dev, bias_dev, y_dev, idx_dev, B, H, W, Cin, Cout, ksize, pad,
               relu, pool);
  return check_launch("aocr_conv2d_forward");
}
int aocr_conv2d_backward_data(void* stream, int32_t compute, const float* dy_dev, const float* w_dev, float* dx_dev, int32_t B, int32_t H,
                              int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad) {
  REQUIRE(dy_dev && w_dev && dx_dev && Cin % 16 == 0 && Cout % 16 == 0, "bad arguments");
  conv_backward_data((hipStream_t)stream, compute == AOCR_COMPUTE_BF16, dy_dev, w_dev, dx_dev, B, H, W, Cin, Cout, ksize, pad);
  return check_launch("aocr_conv2d_backward_data");
}
int aocr_conv2d_backward_filter(void* stream, int32_t compute, const float* x_dev, const float* dy_dev, float* dw_dev, float* dbias_dev,
                                int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize, int32_t pad) {
  REQUIRE(x_dev && dy_dev && dw_dev && Cin % 16 == 0 && Cout % 16 == 0, "bad arguments");
  conv_backward_filter((hipStream_t)stream, compute == AOCR_COMPUTE_BF16, x_dev, dy_dev, dw_dev, dbias_dev, B, H, W, Cin, Cout, ksize, pad);
  return check_launch("aocr_conv2d_backward_filter");
}
int aocr_unpool_relu_backward(void* stream, const float* dpooled_dev, const float* pooled_dev, const uint8_t* idx_dev, float* dy_dev,
                              int32_t B, int32_t Ho, int32_t Wo, int32_t C, int32_t pool) {
  REQUIRE(dpooled_dev && pooled_dev && idx_dev && dy_dev && (pool == 1 || pool == 2) && C % 4 == 0, "bad arguments");
  unpool_relu_backward((hipStream_t)stream, dpooled_dev, pooled_dev, idx_dev, dy_dev, B, Ho, Wo, C, pool);
  return check_launch("aocr_unpool_relu_backward");
}
int aocr_conv1_forward(void* stream, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int32_t B, int32_t H,
                       int32_t W) {
  REQUIRE(x_dev && w_dev && bias_dev && y_dev, "NULL argument");
  conv1_forward((hipStream_t)stream, x_dev, w_dev, bias_dev, y_dev, B, H, W);
  return check_launch("aocr_conv1_forward");
}
int aocr_conv1_backward(void* stream, const float* x_dev, const float* w_dev, const float* bias_dev, const float* dy_pooled_dev,
                        float* dw_dev, float* dbias_dev, int32_t B, int32_t H, int32_t W) {
  REQUIRE(x_dev && w_dev && bias_dev && dy_pooled_dev && dw_dev && dbias_dev, "NULL argument");
  conv1_backward((hipStream_t)stream, x_dev, w_dev, bias_dev, dy_pooled_dev, dw_dev, dbias_dev, B, H, W);
  return check_launch("aocr_conv1_backward");
}
int aocr_batchnorm_relu_forward(void* stream, const float* x_dev, float* y_dev, const float* weight_dev, const float* bias_dev,
                                float* running_mean_dev, float* running_var_dev, float* save_dev, void* scratch_dev, int64_t rows,
                                int32_t C, int32_t training, int32_t update_running, int32_t tb_rows) {
  REQUIRE(x_dev && y_dev && weight_dev && bias_dev && running_mean_dev && running_var_dev && save_dev && scratch_dev, "NULL argument");
  REQUIRE(C % 4 == 0 && C <= 512, "C must be a multiple of 4 and <= 512");
  bn_relu_forward((hipStream_t)stream, x_dev, y_dev, weight_dev, bias_dev, running_mean_dev, running_var_dev, save_dev, scratch_dev, rows, C,
                  training, update_running, tb_rows);
  return check_launch("aocr_batchnorm_relu_forward");
}
int aocr_batchnorm_relu_backward(void* stream, const float* x_dev, const float* y_dev, const float* dA_dev, const float* weight_dev,
                                 const float* save_dev, float* dx_dev, float* dweight_dev, float* dbias_dev, void* scratch_dev,
                                 int64_t rows, int32_t C, int32_t tb_rows) {
  REQUIRE(x_dev && y_dev && dA_dev && weight_dev && save_dev && dx_dev && dweight_dev && dbias_dev && scratch_dev, "NULL argument");
  bn_relu_backward((hipStream_t)stream, x_dev, y_dev, dA_dev, weight_dev, save_dev, dx_dev, dweight_dev, dbias_dev, scratch_dev, rows, C,
                   tb_rows);
  return check_launch("aocr_batchnorm_relu_backward");
}
int aocr_lstm_cell_forward(void* stream, int32_t compute, const float* x_dev, int32_t in_size, const float* h_prev_dev,
                           const float* c_prev_dev, const float* w_i2h_dev, const float* b_i2h_dev, const float* w_h2h_dev,
                           const float* b_h2h_dev, float* c_dev, float* h_dev, float* gates_dev, int32_t B, int32_t H) {
  REQUIRE(x_dev && h_prev_dev && c_prev_dev && w_i2h_dev && b_i2h_dev && w_h2h_dev && b_h2h_dev && c_dev && h_dev, "NULL argument");
  REQUIRE(in_size % 16 == 0 && H % 16 == 0, "in_size and H must be multiples of 16");
  GatesFwdArgs z;
  z.a = make_loadk2(x_dev, in_size, in_size, h_prev_dev, H, H, B);
  z.b = make_loadk2(w_i2h_dev, in_size, in_size, w_h2h_dev, H, H, 4 * H);
  z.K = in_size + H;
  EpGatesFwd& e = z.ep; e.zx = nullptr; e.ldzx = 0; e.b1 = b_i2h_dev; e.b2 = b_h2h_dev; e.c_prev = c_prev_dev; e.ldcp = H;
  e.c_out = c_dev; e.ldc = H; e.h_out = h_dev; e.ldh = H; e.h_out2 = nullptr; e.ldh2 = 0; e.gates = gates_dev; e.ldg = 4 * H; e.M = B; e.H = H;
  launch_small_gates_fwd((hipStream_t)stream, compute == AOCR_COMPUTE_BF16, 1, &z, B, H);
  return check_launch("aocr_lstm_cell_forward");
}
int aocr_lstm_cell_forward_zx(void* stream, int32_t compute, const float* zx_dev, int64_t ldzx, const float* h_prev_dev, const float* c_prev_dev,
                              const float* w_h2h_dev, float* c_dev, float* h_dev, float* gates_dev, int32_t B, int32_t H) {
  REQUIRE(zx_dev && h_prev_dev && c_prev_dev && w_h2h_dev && c_dev && h_dev, "NULL argument");
  REQUIRE(H % 16 == 0 && ldzx >= 4 * (int64_t)H, "H must be a multiple of 16 and ldzx >= 4H");
  GatesFwdArgs z;
  z.a = make_loadk(h_prev_dev, H, B, H);
  z.b = make_loadk(w_h2h_dev, H, 4 * H, H);
  z.K = H;
  EpGatesFwd& e = z.ep; e.zx = zx_dev; e.ldzx = ldzx; e.b1 = nullptr; e.b2 = nullptr; e.c_prev = c_prev_dev; e.ldcp = H;
  e.c_out = c_dev; e.ldc = H; e.h_out = h_dev; e.ldh = H; e.h_out2 = nullptr; e.ldh2 = 0; e.gates = gates_dev; e.ldg = 4 * H; e.M = B; e.H = H;
  launch_small_gates_fwd((hipStream_t)stream, compute == AOCR_COMPUTE_BF16, 1, &z, B, H);
  return check_launch("aocr_lstm_cell_forward_zx");
}
int aocr_lstm_cell_backward(void* stream, const float* dc_dev, const float* dh_dev, const float* gates_dev, const float* c_prev_dev,
                            const float* c_dev, float* dz_dev, float* dc_prev_dev, int32_t B, int32_t H) {
  REQUIRE(dc_dev && dh_dev && gates_dev && c_prev_dev && c_dev && dz_dev && dc_prev_dev, "NULL argument");
  GatesBwdArgs z;
  z.a = make_loadk(dh_dev, H, B, 0); z.b = make_loadmn(dh_dev, H, H, 0); z.K = 0;     // no GEMM part: dh comes in through dh1
  EpGatesBwd& e = z.ep; e.dh1 = dh_dev; e.ld1 = H; e.dh2 = nullptr; e.ld2 = 0; e.dc_in = dc_dev; e.lddc = H; e.gates = gates_dev; e.ldg = 4 * H;
  e.c_prev = c_prev_dev; e.ldcp = H; e.c = c_dev; e.ldcc = H; e.dz = dz_dev; e.lddz = 4 * H; e.dc_out = dc_prev_dev; e.lddco = H; e.M = B; e.H = H;
  launch_small_gates_bwd((hipStream_t)stream, false, 1, &z, B, H);
  return check_launch("aocr_lstm_cell_backward");
}
int aocr_attention_forward(void* stream, const float* ctx_dev, const float* q_dev, float* a_dev, float* c_dev, int64_t ldc, int32_t B,
                           int32_t T, int32_t Hd) {
  REQUIRE(ctx_dev && q_dev && a_dev && c_dev && Hd % 4 == 0 && ldc % 4 == 0, "bad arguments");
  attention_forward((hipStream_t)stream, ctx_dev, q_dev, a_dev, c_dev, ldc, B, T, Hd, 1);
  return check_launch("aocr_attention_forward");
}
int aocr_attention_backward(void* stream, const float* ctx_dev, const float* q_dev, const float* a_dev, const float* dc_dev, int64_t lddc,
                            float* ds_dev, float* dq_dev, int32_t B, int32_t T, int32_t Hd) {
  REQUIRE(ctx_dev && a_dev && dc_dev && ds_dev && dq_dev && Hd % 4 == 0 && lddc % 4 == 0, "bad arguments");
  attention_backward((hipStream_t)stream, ctx_dev, q_dev, a_dev, dc_dev, lddc, ds_dev, dq_dev, B, T, Hd);
  return check_launch("aocr_attention_backward");
}
int aocr_logsoftmax_nll(void* stream, const float* logits_dev, int64_t ld, const int32_t* targets_dev, float* logp_dev, float* dlogits_dev,
                        float* nll_rows_dev, int64_t rows, int32_t V, float grad_scale) {
  REQUIRE(logits_dev && targets_dev && ld >= V, "bad arguments");
  logsoftmax_nll((hipStream_t)stream, logits_dev, ld, targets_dev, 0, 1, (int)rows, logp_dev, dlogits_dev, nll_rows_dev, rows, V, grad_scale);
  return check_launch("aocr_logsoftmax_nll");
}
int aocr_beam_select(void* stream, const float* logp_dev, const int32_t* prev_tok_dev, float* beam_scores_dev, int32_t* tokens_dev,
                     int32_t* parents_dev, int32_t B, int32_t kin, int32_t kout, int32_t V) {
  REQUIRE(logp_dev && beam_scores_dev && tokens_dev && parents_dev && kin >= 1 && kout >= 1 && kout <= kin * V, "bad arguments");
  beam_select((hipStream_t)stream, logp_dev, prev_tok_dev, beam_scores_dev, tokens_dev, parents_dev, B, kin, kout, V);
  return check_launch("aocr_beam_select");
}
int aocr_beam_select_dict(void* stream, const float* logp_dev, const int32_t* prev_tok_dev, float* beam_scores_dev, int32_t* tokens_dev,
                          int32_t* parents_dev, int32_t B, int32_t kin, int32_t kout, int32_t V, const aocr_trie* trie,
                          const int32_t* loc_in_dev, int32_t* loc_out_dev) {
  REQUIRE(logp_dev && beam_scores_dev && tokens_dev && parents_dev && kin >= 1 && kout >= 1 && kout <= kin * V, "bad arguments");
  REQUIRE(trie && loc_out_dev && (loc_in_dev || !prev_tok_dev) && loc_in_dev != loc_out_dev, "trie / node arrays missing or aliased");
  if (check_trie(trie, V)) return 1;
  TrieView tv{(const unsigned long long*)trie->child_mask_dev, trie->child_base_dev, trie->child_dev, loc_in_dev, loc_out_dev};
  beam_select((hipStream_t)stream, logp_dev, prev_tok_dev, beam_scores_dev, tokens_dev, parents_dev, B, kin, kout, V, nullptr, 0, &tv);
  return check_launch("aocr_beam_select_dict");
}
int aocr_edit_distance(void* stream, const int32_t* labels_dev, const int32_t* targets_dev, int32_t B, int32_t L, int32_t* dist_dev,
                       int32_t* target_len_dev) {
  REQUIRE(labels_dev && targets_dev && dist_dev, "NULL argument");
  REQUIRE(B >= 0 && L >= 1 && (size_t)(L + 1) * 256 <= 160 * 1024, "bad sizes: B=%d L=%d", B, L);
  if (B > 0) edit_distance((hipStream_t)stream, labels_dev, targets_dev, B, L, dist_dev, target_len_dev);
  return check_launch("aocr_edit_distance");
}

int aocr_preprocess_lines(void* stream, const uint8_t* src_dev, const aocr_image_desc* desc_dev, int32_t n_images, int32_t out_h,
                          int32_t out_w, float* out_dev) {
  REQUIRE(src_dev && desc_dev && out_dev, "NULL argument");
  REQUIRE(n_images >= 0 && out_h >= 1 && out_w >= 1, "bad sizes: n_images=%d out_h=%d out_w=%d", n_images, out_h, out_w);
  preprocess_lines((hipStream_t)stream, src_dev, desc_dev, n_images, out_h, out_w, out_dev);
  return check_launch("aocr_preprocess_lines");
}

}  // extern "C"
