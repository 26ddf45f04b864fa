// abi_harness.cc -- drives the C ABI of libaocr (include/aocr.h) with NO PyTorch in the process: raw hipMalloc / hipMemcpy, the NULL
// stream, parameters filled in aocr_param_entry order from the counter-based generator, then exactly the call sequence the Lua shim
// (lua/model.lua, the replacement of src/model/model.lua:18-731) makes for one `-phase train` step followed by a `-phase test` step:
//   aocr_model_create -> aocr_train_forward_backward -> aocr_sgd_step -> aocr_decode (greedy) -> aocr_decode (beam 5) -> destroy.
// Between the train step and the update the SAME step is run a second time through the MODULE-level entry points, one call per
// nn.Module of the reference's graphs (cnn.lua:12-45, LSTM.lua:18-162, output_projector.lua:3-8, criterion.lua:3-9) in the order a
// reference-style `cnn_model:forward(x)` / `decoder_clones[t]:forward(...)` / `cnn_model:backward(...)` walks them -- the executed
// twin of lua/aocr_nn.lua (which cannot run here: no Lua toolchain).  Checked against the fused step's taps and the golden logits.
// The expected values come from the fp64 oracle's golden fixture (tests/golden/feed_ld2.npz), handed over as a plain text file by
// tests/test_abi_harness_gpu.py:  `key n v0 v1 ...` per line.  Exit code 0 = every check passed.
//
// TEST INFRASTRUCTURE (built by __graft_entry__.build() through tests/Makefile.harness, run by the -m gpu suite).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>
#include "../include/aocr.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define AOCR_OK(x) do { if ((x) != 0) { fprintf(stderr, "aocr error: %s (%s:%d)\n", aocr_last_error(), __FILE__, __LINE__); return 3; } } while (0)

// ---- counter-based generator (the same arithmetic as oracle_torch.counter_uniform / counter_normal and aocr/synth.py)
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull; uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
static std::vector<double> counter_uniform(uint64_t seed, uint64_t stream, size_t n) {
  const uint64_t base = splitmix64(seed ^ (stream * 0xD1342543DE82EF95ull));
  std::vector<double> u(n);
  for (size_t i = 0; i < n; ++i) u[i] = (double)(splitmix64(base + i) >> 11) * (1.0 / 9007199254740992.0);
  return u;
}
static std::vector<double> counter_normal(uint64_t seed, uint64_t stream, size_t n) {
  std::vector<double> u = counter_uniform(seed, stream, 2 * n), v(n);
  for (size_t i = 0; i < n; ++i) v[i] = std::sqrt(-2.0 * std::log(std::max(u[2 * i], 1e-300))) * std::cos(2.0 * M_PI * u[2 * i + 1]);
  return v;
}

static std::map<std::string, std::vector<double>> read_expected(const char* path) {
  std::map<std::string, std::vector<double>> m; std::ifstream f(path); std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line); std::string key; size_t n; ss >> key >> n; std::vector<double> v(n);
    for (size_t i = 0; i < n; ++i) ss >> v[i];
    m[key] = v;
  }
  return m;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: abi_harness expected.txt\n"); return 64; }
  auto exp = read_expected(argv[1]);
  if (exp.empty()) { fprintf(stderr, "could not read %s\n", argv[1]); return 64; }
  const int B = 2, W = 36, MAXLEN = 5, L = MAXLEN + 1, Lt = 8;
  aocr_config cfg{}; cfg.batch_size = B; cfg.img_h = 32; cfg.max_img_w = W; cfg.enc_hidden = 16; cfg.enc_layers = 1; cfg.dec_layers = 2;
  cfg.vocab = 39; cfg.emb = 20; cfg.input_feed = 1; cfg.max_decoder_l = Lt; cfg.max_beam = 5; cfg.compute = AOCR_COMPUTE_F32;
  if (aocr_version() != AOCR_VERSION) { fprintf(stderr, "header / library version mismatch\n"); return 1; }

  // ---- parameters: stream id = position in the parameter table (= Torch7 getParameters() order); Torch7 init laws [upstream]
  int64_t counts[AOCR_NUM_GROUPS]; AOCR_OK(aocr_param_counts(&cfg, counts));
  int64_t total = 0; for (int g = 0; g < AOCR_NUM_GROUPS; ++g) total += counts[g];
  std::vector<float> params((size_t)total, 0.f);
  std::map<std::string, std::pair<int64_t, int64_t>> where;      // name -> (offset, numel)
  std::map<std::string, int64_t> fan_in;
  for (int idx = 0;; ++idx) {
    char name[64]; int32_t group, ndim; int64_t off, shape[4];
    const int rc = aocr_param_entry(&cfg, idx, name, &group, &off, &ndim, shape);
    if (rc == 1) break;
    if (rc != 0) { fprintf(stderr, "aocr_param_entry: %s\n", aocr_last_error()); return 3; }
    int64_t n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
    const std::string nm = name; where[nm] = {off, n};
    const bool conv_w = ndim == 4, lin_w = ndim == 2 && nm != "dec.lookup";
    if (conv_w) fan_in[nm.substr(0, nm.size() - 2)] = shape[1] * shape[2] * shape[3];    // [Cout][kH][kW][Cin]
    if (lin_w) fan_in[nm.substr(0, nm.size() - 2)] = shape[1];
    std::vector<double> v(n, 0.0);
    if (nm == "dec.lookup") v = counter_normal(910820, idx, n);                           // LookupTable: N(0,1)
    else if (nm.find(".bn") != std::string::npos) { if (nm.back() == 'w') v = counter_uniform(910820, idx, n); }   // BatchNorm weight U(0,1), bias 0
    else {
      const double s = 1.0 / std::sqrt((double)fan_in[nm.substr(0, nm.size() - 2)]);     // weight and bias: U(+-1/sqrt(fan_in))
      v = counter_uniform(910820, idx, n);
      for (auto& x : v) x = (x * 2.0 - 1.0) * s;
    }
    if (conv_w) {                                        // generated in Torch7 layout [Cout][Cin][kH][kW], stored channels-last
      const int64_t Co = shape[0], kh = shape[1], kw = shape[2], Ci = shape[3];
      for (int64_t o = 0; o < Co; ++o) for (int64_t c = 0; c < Ci; ++c) for (int64_t a = 0; a < kh; ++a) for (int64_t b = 0; b < kw; ++b)
        params[off + ((o * kh + a) * kw + b) * Ci + c] = (float)v[((o * Ci + c) * kh + a) * kw + b];
    } else for (int64_t i = 0; i < n; ++i) params[off + i] = (float)v[i];
  }
  std::vector<float> bn(aocr_bn_state_count(), 0.f);
  { size_t o = 0; for (int c : {256, 512, 512}) { for (int i = 0; i < c; ++i) bn[o + c + i] = 1.f; o += 2 * c; } }   // running_var = 1

  // ---- synthetic batch (oracle_torch.synth_batch, data_gen.lua:100-120 layout)
  std::vector<float> img((size_t)B * 32 * W);
  { auto u = counter_uniform(1234, 1000, img.size()); for (size_t i = 0; i < img.size(); ++i) img[i] = (float)std::floor(u[i] * 256.0); }
  std::vector<int32_t> tgt((size_t)B * L, 1), tge((size_t)B * L, 1);
  {
    auto ul = counter_uniform(1234, 1001, B); std::vector<int> lens(B);
    for (int b = 0; b < B; ++b) lens[b] = 2 + (int)std::floor(ul[b] * (MAXLEN - 2 + 1));
    lens[0] = MAXLEN;
    auto uc = counter_uniform(1234, 1002, (size_t)B * MAXLEN);
    for (int b = 0; b < B; ++b) {
      tgt[(size_t)b * L] = 2;
      for (int i = 0; i < lens[b]; ++i) { const int ch = 4 + (int)std::floor(uc[(size_t)b * MAXLEN + i] * 36.0); tgt[(size_t)b * L + 1 + i] = ch; tge[(size_t)b * L + i] = ch; }
      tge[(size_t)b * L + lens[b]] = 3;
    }
  }

  // ---- device buffers: everything the library touches is handed in by the host
  float *d_params, *d_grads, *d_bn, *d_img, *d_scal, *d_scores, *d_gold; int32_t *d_tgt, *d_tge, *d_labels; void* d_ws;
  const size_t ws = aocr_workspace_bytes(&cfg);
  if (ws == 0) { fprintf(stderr, "aocr_workspace_bytes: %s\n", aocr_last_error()); return 3; }
  HIP_OK(hipMalloc(&d_params, total * 4)); HIP_OK(hipMalloc(&d_grads, total * 4)); HIP_OK(hipMalloc(&d_bn, bn.size() * 4));
  HIP_OK(hipMalloc(&d_ws, ws)); HIP_OK(hipMalloc(&d_img, img.size() * 4)); HIP_OK(hipMalloc(&d_tgt, tgt.size() * 4)); HIP_OK(hipMalloc(&d_tge, tge.size() * 4));
  HIP_OK(hipMalloc(&d_scal, 64)); HIP_OK(hipMalloc(&d_scores, B * 4)); HIP_OK(hipMalloc(&d_gold, B * 4)); HIP_OK(hipMalloc(&d_labels, (size_t)B * Lt * 4));
  HIP_OK(hipMemcpy(d_params, params.data(), total * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemset(d_grads, 0, total * 4));
  HIP_OK(hipMemcpy(d_bn, bn.data(), bn.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_tgt, tgt.data(), tgt.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(d_tge, tge.data(), tge.size() * 4, hipMemcpyHostToDevice));

  aocr_model* m = nullptr;
  AOCR_OK(aocr_model_create(&cfg, d_params, d_grads, d_bn, d_ws, ws, nullptr, &m));
  int failures = 0;
  auto check = [&](const char* what, double got, double want, double tol) {
    const bool ok = std::fabs(got - want) <= tol;
    if (!ok) { ++failures; printf("[harness] FAIL %s: got %.9g want %.9g (tol %.1e)\n", what, got, want, tol); }
    return ok;
  };

  // ---- -phase train: feval (model.lua:284-696) + optim.sgd_list (optim_sgd.lua:38-95)
  AOCR_OK(aocr_train_forward_backward(m, d_img, d_tgt, d_tge, B, W, L, 1.0f / B, d_scal));
  float loss = 0; HIP_OK(hipMemcpy(&loss, d_scal, 4, hipMemcpyDeviceToHost));
  check("train loss (sum over the batch)", loss, exp["loss"][0] * B, 1e-3);
  { const void* p; int32_t nd; int64_t sh[4]; AOCR_OK(aocr_get_tensor(m, "logits", &p, &nd, sh));
    std::vector<float> lg((size_t)sh[0] * sh[1] * sh[2]); HIP_OK(hipMemcpy(lg.data(), p, lg.size() * 4, hipMemcpyDeviceToHost));
    const std::vector<double>& e = exp["logits"]; double worst = 0;
    for (int t = 0; t < L; ++t) for (int b = 0; b < B; ++b) for (int v = 0; v < 39; ++v)
      worst = std::max(worst, std::fabs((double)lg[((size_t)t * sh[1] + b) * sh[2] + v] - e[((size_t)t * B + b) * 39 + v]));
    printf("[harness] decoder logits max-abs error vs the fp64 oracle: %.3e\n", worst);
    check("logits max-abs", worst, 0.0, 1e-4); }
  { // two gradient tensors, all entries (Linear layout = Torch7 layout)
    for (const char* k : {"dec.attn.wa", "proj.w"}) {
      const auto w = where[k]; std::vector<float> g(w.second); HIP_OK(hipMemcpy(g.data(), d_grads + w.first, w.second * 4, hipMemcpyDeviceToHost));
      const std::vector<double>& e = exp[std::string("g:") + k]; double worst = 0, mag = 0;
      for (size_t i = 0; i < e.size(); ++i) { worst = std::max(worst, std::fabs((double)g[i] - e[i])); mag = std::max(mag, std::fabs(e[i])); }
      printf("[harness] grad %s: max-abs error %.3e (max %.3e)\n", k, worst, mag);
      check(k, worst / (mag + 1e-30), 0.0, 2e-3);
    } }
  // =====================================================================================================================
  // module-level twin of lua/aocr_nn.lua: the same step, one ABI call per nn.Module
  // =====================================================================================================================
  {
    const int cmp = AOCR_COMPUTE_F32, He = 16, Hd = 32, E = 20, V = 39, T = W / 4 - 1;
    std::vector<void*> owned;
    auto dalloc = [&](size_t floats) -> float* { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(floats, 4) * 4) != hipSuccess) return nullptr; hipMemset(p, 0, std::max<size_t>(floats, 4) * 4); owned.push_back(p); return (float*)p; };
    auto P = [&](const char* k) -> float* { return d_params + where[k].first; };
    auto tap = [&](const char* name, std::vector<float>& out) -> int {
      const void* p; int32_t nd; int64_t sh[4]; if (aocr_get_tensor(m, name, &p, &nd, sh) != 0) return 1;
      size_t n = 1; for (int i = 0; i < nd; ++i) n *= (size_t)sh[i];
      out.resize(n); return hipMemcpy(out.data(), p, n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1; };
    auto maxdiff = [&](const float* dev, const std::vector<float>& ref, double* mag) -> double {
      std::vector<float> h(ref.size()); hipMemcpy(h.data(), dev, ref.size() * 4, hipMemcpyDeviceToHost);
      double w = 0, g = 0; for (size_t i = 0; i < ref.size(); ++i) { w = std::max(w, std::fabs((double)h[i] - ref[i])); g = std::max(g, std::fabs((double)ref[i])); }
      if (mag) *mag = g; return w; };
    std::vector<float> t_feats, t_ctx, t_logits, t_dfeats;
    if (tap("feats", t_feats) || tap("context", t_ctx) || tap("logits", t_logits) || tap("dfeats", t_dfeats)) { fprintf(stderr, "tap failed: %s\n", aocr_last_error()); return 3; }
    std::vector<float> fused_grads((size_t)total); HIP_OK(hipMemcpy(fused_grads.data(), d_grads, total * 4, hipMemcpyDeviceToHost));
    // geometry of cnn.lua:12-45 on (B,1,32,W)
    const int H1 = 16, W1 = W / 2, H2 = 8, W2 = W1 / 2, H4 = 4, H6 = 2;
    float* bn2 = dalloc(bn.size()); HIP_OK(hipMemcpy(bn2, bn.data(), bn.size() * 4, hipMemcpyHostToDevice));   // a scratch copy: the fused step already updated the real running statistics
    float* scratch = dalloc(AOCR_BN_SCRATCH_BYTES / 4);
    float *A1 = dalloc((size_t)B * H1 * W1 * 64), *A2 = dalloc((size_t)B * H2 * W2 * 128), *Y3 = dalloc((size_t)B * H2 * W2 * 256), *A3 = dalloc((size_t)B * H2 * W2 * 256);
    float *A4 = dalloc((size_t)B * H4 * W2 * 256), *Y5 = dalloc((size_t)B * H4 * W2 * 512), *A5 = dalloc((size_t)B * H4 * W2 * 512), *A6 = dalloc((size_t)B * H6 * W2 * 512);
    float *Y7 = dalloc((size_t)B * T * 512), *X = dalloc((size_t)T * B * 512), *sv3 = dalloc(512), *sv5 = dalloc(1024), *sv7 = dalloc(1024);
    uint8_t *i2 = (uint8_t*)dalloc((size_t)B * H2 * W2 * 128), *i4 = (uint8_t*)dalloc((size_t)B * H4 * W2 * 256), *i6 = (uint8_t*)dalloc((size_t)B * H6 * W2 * 512);
    // ---- createCNNModel, cnn.lua:9-45: AddConstant/MulConstant + conv1 + ReLU + pool are ONE call; conv + ReLU + pool fuse; BN + ReLU fuse
    AOCR_OK(aocr_conv1_forward(nullptr, d_img, P("cnn.conv1.w"), P("cnn.conv1.b"), A1, B, 32, W));
    AOCR_OK(aocr_conv2d_forward(nullptr, cmp, A1, P("cnn.conv2.w"), P("cnn.conv2.b"), A2, i2, B, H1, W1, 64, 128, 3, 1, 1, 1));
    AOCR_OK(aocr_conv2d_forward(nullptr, cmp, A2, P("cnn.conv3.w"), P("cnn.conv3.b"), Y3, nullptr, B, H2, W2, 128, 256, 3, 1, 0, 0));
    AOCR_OK(aocr_batchnorm_relu_forward(nullptr, Y3, A3, P("cnn.bn3.w"), P("cnn.bn3.b"), bn2, bn2 + 256, sv3, scratch, (int64_t)B * H2 * W2, 256, 1, 1, 0));
    AOCR_OK(aocr_conv2d_forward(nullptr, cmp, A3, P("cnn.conv4.w"), P("cnn.conv4.b"), A4, i4, B, H2, W2, 256, 256, 3, 1, 1, 2));
    AOCR_OK(aocr_conv2d_forward(nullptr, cmp, A4, P("cnn.conv5.w"), P("cnn.conv5.b"), Y5, nullptr, B, H4, W2, 256, 512, 3, 1, 0, 0));
    AOCR_OK(aocr_batchnorm_relu_forward(nullptr, Y5, A5, P("cnn.bn5.w"), P("cnn.bn5.b"), bn2 + 512, bn2 + 1024, sv5, scratch, (int64_t)B * H4 * W2, 512, 1, 1, 0));
    AOCR_OK(aocr_conv2d_forward(nullptr, cmp, A5, P("cnn.conv6.w"), P("cnn.conv6.b"), A6, i6, B, H4, W2, 512, 512, 3, 1, 1, 2));
    AOCR_OK(aocr_conv2d_forward(nullptr, cmp, A6, P("cnn.conv7.w"), P("cnn.conv7.b"), Y7, nullptr, B, H6, W2, 512, 512, 2, 0, 0, 0));
    AOCR_OK(aocr_batchnorm_relu_forward(nullptr, Y7, X, P("cnn.bn7.w"), P("cnn.bn7.b"), bn2 + 1536, bn2 + 2048, sv7, scratch, (int64_t)B * T, 512, 1, 1, B));   // + View / Transpose -> (T,B,512)
    { double mag; const double e = maxdiff(X, t_feats, &mag); printf("[harness] module-level CNN output vs the fused step: max-abs %.3e (max %.3e)\n", e, mag); check("modules: feats", e, 0.0, 1e-5 * std::max(1.0, mag)); }
    { std::vector<float> b2(bn.size()), b1(bn.size()); HIP_OK(hipMemcpy(b2.data(), bn2, bn.size() * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(b1.data(), d_bn, bn.size() * 4, hipMemcpyDeviceToHost));
      double w = 0; for (size_t i = 0; i < bn.size(); ++i) w = std::max(w, std::fabs((double)b2[i] - b1[i])); check("modules: BatchNorm running statistics", w, 0.0, 1e-5); }
    // ---- encoder, model.lua:291-316: one cell call per (direction, time step); context (B,T,2He)
    float* ctx = dalloc((size_t)B * T * Hd); float* zero = dalloc((size_t)B * Hd);
    float *ec[2][2], *eh[2][2], *eg = dalloc((size_t)B * 4 * He);
    for (int d = 0; d < 2; ++d) for (int k = 0; k < 2; ++k) { ec[d][k] = dalloc((size_t)B * He); eh[d][k] = dalloc((size_t)B * He); }
    int efin[2] = {0, 0};
    for (int d = 0; d < 2; ++d) {
      const std::string pf = d == 0 ? "enc_fw.l1." : "enc_bw.l1.";
      int cur = 0;
      for (int step = 0; step < T; ++step) {
        const int t = d == 0 ? step : T - 1 - step;
        AOCR_OK(aocr_lstm_cell_forward(nullptr, cmp, X + (size_t)t * B * 512, 512, eh[d][cur], ec[d][cur], P((pf + "i2h.w").c_str()), P((pf + "i2h.b").c_str()),
                                       P((pf + "h2h.w").c_str()), P((pf + "h2h.b").c_str()), ec[d][cur ^ 1], eh[d][cur ^ 1], eg, B, He));
        cur ^= 1;
        HIP_OK(hipMemcpy2D(ctx + (size_t)t * Hd + d * He, (size_t)T * Hd * 4, eh[d][cur], He * 4, He * 4, B, hipMemcpyDeviceToDevice));     // context[{{},t,{dir}}]:copy(h), model.lua:303,315
      }
      efin[d] = cur;
    }
    { double mag; const double e = maxdiff(ctx, t_ctx, &mag); printf("[harness] module-level context vs the fused step: max-abs %.3e\n", e); check("modules: context", e, 0.0, 1e-5); }
    // ---- decoder clones, model.lua:537-569 + LSTM.lua:18-162: LookupTable, two LSTM layers, attention, combine; projector + criterion
    float *dc[2][2], *dh[2][2];
    for (int l = 0; l < 2; ++l) for (int k = 0; k < 2; ++k) { dc[l][k] = dalloc((size_t)B * Hd); dh[l][k] = dalloc((size_t)B * Hd); }
    for (int d = 0; d < 2; ++d) {     // layer-1 state = final encoder states, forward | backward halves (model.lua:543-548); quirk S5: h1(0) stays zero
      HIP_OK(hipMemcpy2D(dc[0][0] + d * He, Hd * 4, ec[d][efin[d]], He * 4, He * 4, B, hipMemcpyDeviceToDevice));
    }
    float *emb = dalloc((size_t)B * E), *zx = dalloc((size_t)B * 4 * Hd), *feed = dalloc((size_t)B * Hd), *q = dalloc((size_t)B * Hd), *att = dalloc((size_t)B * T);
    float *cat = dalloc((size_t)B * 2 * Hd), *gates = dalloc((size_t)B * 4 * Hd), *logits = dalloc((size_t)L * B * 40), *nll = dalloc((size_t)L * B), *dlog = dalloc((size_t)L * B * 40);
    float *outs = dalloc((size_t)L * B * Hd), *bsum1 = dalloc(4 * Hd);
    AOCR_OK(aocr_pointwise(nullptr, AOCR_PW_ADD, P("dec.l1.i2h.b"), P("dec.l1.h2h.b"), bsum1, 4 * Hd));
    int32_t* ids; HIP_OK(hipMalloc(&ids, (size_t)L * B * 4)); owned.push_back(ids);
    int32_t* tge_tm; HIP_OK(hipMalloc(&tge_tm, (size_t)L * B * 4)); owned.push_back(tge_tm);
    { std::vector<int32_t> a((size_t)L * B), e2((size_t)L * B); for (int t = 0; t < L; ++t) for (int b = 0; b < B; ++b) { a[(size_t)t * B + b] = tgt[(size_t)b * L + t]; e2[(size_t)t * B + b] = tge[(size_t)b * L + t]; }
      HIP_OK(hipMemcpy(ids, a.data(), a.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(tge_tm, e2.data(), e2.size() * 4, hipMemcpyHostToDevice)); }
    int cur = 0;
    for (int t = 0; t < L; ++t) {
      AOCR_OK(aocr_lookup_forward(nullptr, P("dec.lookup"), ids + (size_t)t * B, emb, B, E));                                       // nn.LookupTable, LSTM.lua:55-56
      // layer 1: i2h over JoinTable{embedding, input feed} (LSTM.lua:59-64,79-80) as two products into one pre-activation, then the cell
      AOCR_OK(aocr_gemm(nullptr, cmp, emb, E, 1, P("dec.l1.i2h.w"), E + Hd, 1, zx, 4 * Hd, B, 4 * Hd, E, bsum1, 0));
      AOCR_OK(aocr_gemm(nullptr, cmp, t == 0 ? zero : feed, Hd, 1, P("dec.l1.i2h.w") + E, E + Hd, 1, zx, 4 * Hd, B, 4 * Hd, Hd, nullptr, AOCR_GEMM_ACCUMULATE));
      AOCR_OK(aocr_lstm_cell_forward_zx(nullptr, cmp, zx, 4 * Hd, dh[0][cur], dc[0][cur], P("dec.l1.h2h.w"), dc[0][cur ^ 1], dh[0][cur ^ 1], gates, B, Hd));
      AOCR_OK(aocr_lstm_cell_forward(nullptr, cmp, dh[0][cur ^ 1], Hd, dh[1][cur], dc[1][cur], P("dec.l2.i2h.w"), P("dec.l2.i2h.b"), P("dec.l2.h2h.w"), P("dec.l2.h2h.b"),
                                     dc[1][cur ^ 1], dh[1][cur ^ 1], gates, B, Hd));
      cur ^= 1;
      // create_decoder_attn, LSTM.lua:124-162: q = W_a h (LinearNoBias), scores / softmax / context, JoinTable{c, h}, LinearNoBias + Tanh
      AOCR_OK(aocr_gemm(nullptr, cmp, dh[1][cur], Hd, 1, P("dec.attn.wa"), Hd, 1, q, Hd, B, Hd, Hd, nullptr, 0));
      AOCR_OK(aocr_attention_forward(nullptr, ctx, q, att, cat, 2 * Hd, B, T, Hd));                                                     // c lands in the first half of the JoinTable buffer
      HIP_OK(hipMemcpy2D(cat + Hd, 2 * Hd * 4, dh[1][cur], Hd * 4, Hd * 4, B, hipMemcpyDeviceToDevice));
      AOCR_OK(aocr_gemm(nullptr, cmp, cat, 2 * Hd, 1, P("dec.attn.wc"), 2 * Hd, 1, outs + (size_t)t * B * Hd, Hd, B, Hd, 2 * Hd, nullptr, AOCR_GEMM_TANH));
      HIP_OK(hipMemcpy(feed, outs + (size_t)t * B * Hd, (size_t)B * Hd * 4, hipMemcpyDeviceToDevice));                                   // input feed of the next clone
      // createOutputUnit, output_projector.lua:3-8 (the LogSoftMax is fused with the criterion below)
      AOCR_OK(aocr_gemm(nullptr, cmp, outs + (size_t)t * B * Hd, Hd, 1, P("proj.w"), Hd, 1, logits + (size_t)t * B * 40, 40, B, V, Hd, P("proj.b"), 0));
    }
    // criterion.lua:3-9 + model.lua:644-648: ClassNLLCriterion(weights, PAD weight 0, sizeAverage false), d(loss) scaled by 1 / batch_size
    AOCR_OK(aocr_logsoftmax_nll(nullptr, logits, 40, tge_tm, nullptr, dlog, nll, (int64_t)L * B, V, 1.0f / B));
    { std::vector<float> lg((size_t)L * B * 40), nl((size_t)L * B); HIP_OK(hipMemcpy(lg.data(), logits, lg.size() * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(nl.data(), nll, nl.size() * 4, hipMemcpyDeviceToHost));
      const std::vector<double>& e = exp["logits"]; double worst = 0, wf = 0, ls = 0;
      for (int t = 0; t < L; ++t) for (int b = 0; b < B; ++b) for (int v = 0; v < V; ++v) {
        worst = std::max(worst, std::fabs((double)lg[((size_t)t * B + b) * 40 + v] - e[((size_t)t * B + b) * V + v]));
        wf = std::max(wf, std::fabs((double)lg[((size_t)t * B + b) * 40 + v] - t_logits[((size_t)t * B + b) * 40 + v])); }
      for (float x : nl) ls += x;
      printf("[harness] module-level decoder logits: max-abs %.3e vs the fp64 oracle, %.3e vs the fused step; loss %.6f\n", worst, wf, ls);
      check("modules: logits vs oracle", worst, 0.0, 1e-4); check("modules: logits vs fused", wf, 0.0, 2e-5); check("modules: loss", ls, exp["loss"][0] * B, 1e-3); }
    // projector backward (model.lua:648): gradWeight = dlogits^T out, gradBias = column sums -- against the fused step's gradients
    { float* dwo = dalloc((size_t)V * Hd);
      AOCR_OK(aocr_gemm(nullptr, cmp, dlog, 40, 0, outs, Hd, 0, dwo, Hd, V, Hd, L * B, nullptr, 0));
      const auto w = where["proj.w"]; std::vector<float> ref(fused_grads.begin() + w.first, fused_grads.begin() + w.first + w.second);
      double mag; const double e = maxdiff(dwo, ref, &mag); printf("[harness] module-level proj.w gradient vs the fused step: max-abs %.3e (max %.3e)\n", e, mag);
      check("modules: d proj.w", e / (mag + 1e-30), 0.0, 1e-4); }
    // ---- cnn_model:backward(cnn_input, cnn_grad), model.lua:692, module by module in reverse (d(features) from the fused step's tap)
    { float* dX = dalloc(t_dfeats.size()); HIP_OK(hipMemcpy(dX, t_dfeats.data(), t_dfeats.size() * 4, hipMemcpyHostToDevice));
      const size_t gmax = (size_t)B * H1 * W1 * 128;
      float *G0 = dalloc(gmax), *G1 = dalloc(gmax), *mg = dalloc((size_t)total);     // mg: module-level gradient vector, same layout as d_grads
      auto Gp = [&](const char* k) -> float* { return mg + where[k].first; };
      AOCR_OK(aocr_batchnorm_relu_backward(nullptr, Y7, X, dX, P("cnn.bn7.w"), sv7, G0, Gp("cnn.bn7.w"), Gp("cnn.bn7.b"), scratch, (int64_t)B * T, 512, B));
      AOCR_OK(aocr_conv2d_backward_filter(nullptr, cmp, A6, G0, Gp("cnn.conv7.w"), Gp("cnn.conv7.b"), B, H6, W2, 512, 512, 2, 0));
      AOCR_OK(aocr_conv2d_backward_data(nullptr, cmp, G0, P("cnn.conv7.w"), G1, B, H6, W2, 512, 512, 2, 0));
      AOCR_OK(aocr_unpool_relu_backward(nullptr, G1, A6, i6, G0, B, H4, W2, 512, 2));
      AOCR_OK(aocr_conv2d_backward_filter(nullptr, cmp, A5, G0, Gp("cnn.conv6.w"), Gp("cnn.conv6.b"), B, H4, W2, 512, 512, 3, 1));
      AOCR_OK(aocr_conv2d_backward_data(nullptr, cmp, G0, P("cnn.conv6.w"), G1, B, H4, W2, 512, 512, 3, 1));
      AOCR_OK(aocr_batchnorm_relu_backward(nullptr, Y5, A5, G1, P("cnn.bn5.w"), sv5, G0, Gp("cnn.bn5.w"), Gp("cnn.bn5.b"), scratch, (int64_t)B * H4 * W2, 512, 0));
      AOCR_OK(aocr_conv2d_backward_filter(nullptr, cmp, A4, G0, Gp("cnn.conv5.w"), Gp("cnn.conv5.b"), B, H4, W2, 256, 512, 3, 1));
      AOCR_OK(aocr_conv2d_backward_data(nullptr, cmp, G0, P("cnn.conv5.w"), G1, B, H4, W2, 256, 512, 3, 1));
      AOCR_OK(aocr_unpool_relu_backward(nullptr, G1, A4, i4, G0, B, H2, W2, 256, 2));
      AOCR_OK(aocr_conv2d_backward_filter(nullptr, cmp, A3, G0, Gp("cnn.conv4.w"), Gp("cnn.conv4.b"), B, H2, W2, 256, 256, 3, 1));
      AOCR_OK(aocr_conv2d_backward_data(nullptr, cmp, G0, P("cnn.conv4.w"), G1, B, H2, W2, 256, 256, 3, 1));
      AOCR_OK(aocr_batchnorm_relu_backward(nullptr, Y3, A3, G1, P("cnn.bn3.w"), sv3, G0, Gp("cnn.bn3.w"), Gp("cnn.bn3.b"), scratch, (int64_t)B * H2 * W2, 256, 0));
      AOCR_OK(aocr_conv2d_backward_filter(nullptr, cmp, A2, G0, Gp("cnn.conv3.w"), Gp("cnn.conv3.b"), B, H2, W2, 128, 256, 3, 1));
      AOCR_OK(aocr_conv2d_backward_data(nullptr, cmp, G0, P("cnn.conv3.w"), G1, B, H2, W2, 128, 256, 3, 1));
      AOCR_OK(aocr_unpool_relu_backward(nullptr, G1, A2, i2, G0, B, H1, W1, 128, 1));
      AOCR_OK(aocr_conv2d_backward_filter(nullptr, cmp, A1, G0, Gp("cnn.conv2.w"), Gp("cnn.conv2.b"), B, H1, W1, 64, 128, 3, 1));
      AOCR_OK(aocr_conv2d_backward_data(nullptr, cmp, G0, P("cnn.conv2.w"), G1, B, H1, W1, 64, 128, 3, 1));
      AOCR_OK(aocr_conv1_backward(nullptr, d_img, P("cnn.conv1.w"), P("cnn.conv1.b"), G1, Gp("cnn.conv1.w"), Gp("cnn.conv1.b"), B, 32, W));
      double worst = 0; std::string wk;
      for (const char* k : {"cnn.conv1.w", "cnn.conv1.b", "cnn.conv2.w", "cnn.conv2.b", "cnn.conv3.w", "cnn.bn3.w", "cnn.bn3.b", "cnn.conv4.w", "cnn.conv4.b", "cnn.conv5.w",
                            "cnn.bn5.w", "cnn.bn5.b", "cnn.conv6.w", "cnn.conv6.b", "cnn.conv7.w", "cnn.bn7.w", "cnn.bn7.b"}) {      // (conv3/5/7 biases sit in front of a BatchNorm: exact gradient 0, rounding noise)
        const auto w = where[k]; std::vector<float> ref(fused_grads.begin() + w.first, fused_grads.begin() + w.first + w.second);
        double mag; const double e = maxdiff(Gp(k), ref, &mag) / (mag + 1e-30);
        if (e > worst) { worst = e; wk = k; }
        check((std::string("modules: d ") + k).c_str(), e, 0.0, 2e-4);
      }
      printf("[harness] module-level CNN backward vs the fused step, 17 gradient tensors: worst relative max-abs %.3e (%s)\n", worst, wk.c_str()); }
    HIP_OK(hipDeviceSynchronize());
    for (void* p : owned) hipFree(p);
  }
  AOCR_OK(aocr_sgd_step(m, 0.1f, 5.0f, d_scal + 2));
  { float n[10]; HIP_OK(hipMemcpy(n, d_scal + 2, 40, hipMemcpyDeviceToHost));
    for (int g = 0; g < 5; ++g) { check("param norm", n[2 * g], exp["norms"][2 * g], 1e-4 * std::max(1.0, exp["norms"][2 * g])); check("grad norm", n[2 * g + 1], exp["norms"][2 * g + 1], 2e-3 * std::max(1e-3, exp["norms"][2 * g + 1])); } }

  // ---- -phase test on the updated parameters / running statistics: greedy, then beam 5 (model.lua:321-627)
  for (int beam : {1, 5}) {
    AOCR_OK(aocr_decode(m, d_img, d_tgt, d_tge, B, W, L, beam, d_labels, d_scores, d_gold, d_scal));
    std::vector<int32_t> lab((size_t)B * Lt); float sc[B], gd[B], ls;
    HIP_OK(hipMemcpy(lab.data(), d_labels, lab.size() * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(sc, d_scores, B * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(gd, d_gold, B * 4, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(&ls, d_scal, 4, hipMemcpyDeviceToHost));
    const std::string p = "dec" + std::to_string(beam) + ":";
    for (size_t i = 0; i < lab.size(); ++i) check((p + "labels").c_str(), lab[i], exp[p + "labels"][i], 0.0);
    for (int b = 0; b < B; ++b) { check((p + "scores").c_str(), sc[b], exp[p + "scores"][b], 2e-3); check((p + "gold").c_str(), gd[b], exp[p + "gold"][b], 2e-3); }
    check((p + "loss").c_str(), ls, exp[p + "loss"][0], 2e-3 * std::max(1.0, exp[p + "loss"][0]));
    printf("[harness] beam %d: labels[0] =", beam); for (int t = 0; t < Lt; ++t) printf(" %d", lab[t]); printf("  score %.5f gold %.5f loss %.5f\n", sc[0], gd[0], ls);
  }
  AOCR_OK(aocr_model_destroy(m));
  hipFree(d_params); hipFree(d_grads); hipFree(d_bn); hipFree(d_ws); hipFree(d_img); hipFree(d_tgt); hipFree(d_tge); hipFree(d_scal); hipFree(d_scores);
  hipFree(d_gold); hipFree(d_labels);
  printf("[harness] %s (%d failed checks)\n", failures ? "FAILED" : "PASSED", failures);
  return failures ? 1 : 0;
}
