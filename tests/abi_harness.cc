// abi_harness.cc -- drives the C ABI of libaocr (include/aocr.h) with NO PyTorch in the process: raw hipMalloc / hipMemcpy, the NULL
// stream, parameters filled in aocr_param_entry order from the counter-based generator, then exactly the call sequence the Lua shim
// (lua/model.lua, the replacement of src/model/model.lua:18-731) makes for one `-phase train` step followed by a `-phase test` step:
//   aocr_model_create -> aocr_train_forward_backward -> aocr_sgd_step -> aocr_decode (greedy) -> aocr_decode (beam 5) -> destroy.
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
