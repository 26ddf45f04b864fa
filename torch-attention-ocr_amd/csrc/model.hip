// model.hip -- fused sequence-level drivers: what Model:step's feval does (src/model/model.lua:284-696),
// as straight-line kernel launches on one HIP stream with a native time loop in place of the reference's
// cloned cells (model_utils.lua:3-50).  Weight gradients of every recurrent Linear are hoisted out of the
// time loops into one large-K contraction each; the loops keep only what is truly sequential.
#include "model.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace aocr {

// ------------------------------------------------------------------------------------------------
// parameter layout: Torch7 getParameters() order per group (module order, weight then bias), model.lua:150,163-168
// ------------------------------------------------------------------------------------------------
static void add_entry(Layout& L, int64_t& off, const std::string& name, int group, std::initializer_list<int64_t> shape) {
  ParamEntry e; e.name = name; e.group = group; e.offset = off; e.ndim = (int)shape.size(); e.numel = 1;
  int i = 0; for (int64_t v : shape) { e.shape[i++] = v; e.numel *= v; }
  for (; i < 4; ++i) e.shape[i] = 1;
  off += e.numel; L.e.push_back(e);
}
Layout build_layout(const aocr_config& c) {
  Layout L; int64_t off = 0;
  const int He = c.enc_hidden, Hd = 2 * He;
  static const int convs[7][4] = {{1, 64, 3, 1}, {64, 128, 3, 1}, {128, 256, 3, 1}, {256, 256, 3, 1},
                                  {256, 512, 3, 1}, {512, 512, 3, 1}, {512, 512, 2, 0}};
  L.group_off[0] = 0;
  for (int i = 0; i < 7; ++i) {                                            // cnn.lua:12-42
    char nm[64];
    snprintf(nm, 64, "cnn.conv%d.w", i + 1); add_entry(L, off, nm, 0, {convs[i][1], convs[i][2], convs[i][2], convs[i][0]});
    snprintf(nm, 64, "cnn.conv%d.b", i + 1); add_entry(L, off, nm, 0, {convs[i][1]});
    if (i == 2 || i == 4 || i == 6) {
      snprintf(nm, 64, "cnn.bn%d.w", i + 1); add_entry(L, off, nm, 0, {convs[i][1]});
      snprintf(nm, 64, "cnn.bn%d.b", i + 1); add_entry(L, off, nm, 0, {convs[i][1]});
    }
  }
  L.group_off[1] = off;
  auto lstm = [&](const char* prefix, int group, int in0, int H, int layers) {
    for (int l = 1; l <= layers; ++l) {
      int in = l == 1 ? in0 : H; char nm[64];
      snprintf(nm, 64, "%s.l%d.i2h.w", prefix, l); add_entry(L, off, nm, group, {4 * H, in});
      snprintf(nm, 64, "%s.l%d.i2h.b", prefix, l); add_entry(L, off, nm, group, {4 * H});
      snprintf(nm, 64, "%s.l%d.h2h.w", prefix, l); add_entry(L, off, nm, group, {4 * H, H});
      snprintf(nm, 64, "%s.l%d.h2h.b", prefix, l); add_entry(L, off, nm, group, {4 * H});
    }
  };
  lstm("enc_fw", 1, 512, He, c.enc_layers); L.group_off[2] = off;
  lstm("enc_bw", 2, 512, He, c.enc_layers); L.group_off[3] = off;
  add_entry(L, off, "dec.lookup", 3, {c.vocab, c.emb});
  lstm("dec", 3, c.emb + (c.input_feed ? Hd : 0), Hd, c.dec_layers);
  add_entry(L, off, "dec.attn.wa", 3, {Hd, Hd});
  add_entry(L, off, "dec.attn.wc", 3, {Hd, 2 * Hd});
  L.group_off[4] = off;
  add_entry(L, off, "proj.w", 4, {c.vocab, Hd});
  add_entry(L, off, "proj.b", 4, {c.vocab});
  L.group_off[5] = off;
  return L;
}

bool make_dims(const aocr_config& c, int B, int W, int L, Dims& d) {
  d.B = B; d.H = c.img_h; d.W = W; d.L = L;
  d.H1 = d.H / 2; d.W1 = W / 2;               // after pool 1
  d.H2 = d.H1 / 2; d.W2 = d.W1 / 2;           // after pool 2 (width stays W2 from here)
  d.H4 = d.H2 / 2; d.H6 = d.H4 / 2;           // after the two (2,1) pools
  d.Ho7 = d.H6 - 1; d.Wo7 = d.W2 - 1;         // conv7: 2x2, pad 0
  d.T = d.Ho7 * d.Wo7;                        // View(512,-1), cnn.lua:44
  return B >= 1 && d.Ho7 >= 1 && d.Wo7 >= 1 && L >= 1;
}

// per-family profile marks (aocr_profile_enable): one timing event per change of family
void prof_mark_slow(aocr_model* m, int tag) {
  if (m->prof_n > 0 && m->prof_tag[m->prof_n - 1] == tag) return;          // same family continues
  if (m->prof_n == m->prof_ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; m->prof_ev.push_back(e); m->prof_tag.push_back(0); }
  m->prof_tag[m->prof_n] = tag;
  hipEventRecord(m->prof_ev[m->prof_n], m->s);
  ++m->prof_n;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
int model_carve(aocr_model* m, void* base, size_t bytes) {
  const aocr_config& c = m->cfg;
  Dims d; if (!make_dims(c, c.batch_size, c.max_img_w, c.max_decoder_l, d)) return -1;
  Arena a; a.base = (char*)base; a.off = 0;
  const size_t B = d.B, T = d.T, L = d.L, He = m->He, Hd = m->Hd, E = m->E;
  const size_t R = B * (size_t)(c.max_beam > 1 ? c.max_beam : 1);
  m->A1 = a.get<float>(B * d.H1 * d.W1 * 64);
  m->A2 = a.get<float>(B * d.H2 * d.W2 * 128); m->idx2 = a.get<uint8_t>(B * d.H2 * d.W2 * 128);
  m->Y3 = a.get<float>(B * d.H2 * d.W2 * 256); m->A3 = a.get<float>(B * d.H2 * d.W2 * 256);
  m->A4 = a.get<float>(B * d.H4 * d.W2 * 256); m->idx4 = a.get<uint8_t>(B * d.H4 * d.W2 * 256);
  m->Y5 = a.get<float>(B * d.H4 * d.W2 * 512); m->A5 = a.get<float>(B * d.H4 * d.W2 * 512);
  m->A6 = a.get<float>(B * d.H6 * d.W2 * 512); m->idx6 = a.get<uint8_t>(B * d.H6 * d.W2 * 512);
  m->Y7 = a.get<float>(B * T * 512); m->X = a.get<float>(T * B * 512); m->dX = a.get<float>(T * B * 512);
  size_t gmax = B * d.H1 * d.W1 * 128;                                      // d(conv2 pre-pool output): the largest gradient map
  m->G0 = a.get<float>(gmax); m->G1 = a.get<float>(gmax); m->gmax = gmax;
  m->A1b = m->A2b = m->A3b = m->A4b = m->A5b = m->A6b = m->G0b = m->G2b = nullptr;
  m->route1 = a.get<uint16_t>(conv1_route_elems((int)B, d.H, d.W)); m->route1_valid = false;
  for (int i = 0; i < 8; ++i) { m->wb[i] = nullptr; m->wtb[i] = nullptr; m->wtf[i] = nullptr; }
  if (!m->bf16) {
    static const int wsz32[8] = {0, 0, 128 * 9 * 64, 256 * 9 * 128, 256 * 9 * 256, 512 * 9 * 256, 512 * 9 * 512, 512 * 4 * 512};
    for (int i = 2; i <= 7; ++i) m->wtf[i] = a.get<float>(wsz32[i]);
  }
  if (m->bf16) {
    m->A1b = a.get<bf16_t>(B * d.H1 * d.W1 * 64); m->A2b = a.get<bf16_t>(B * d.H2 * d.W2 * 128);
    m->A3b = a.get<bf16_t>(B * d.H2 * d.W2 * 256); m->A4b = a.get<bf16_t>(B * d.H4 * d.W2 * 256);
    m->A5b = a.get<bf16_t>(B * d.H4 * d.W2 * 512); m->A6b = a.get<bf16_t>(B * d.H6 * d.W2 * 512);
    m->G0b = a.get<bf16_t>(gmax); m->G2b = a.get<bf16_t>(gmax);      // G2b: second gradient-map shadow (cnn_backward: filter gradients on the side stream)
    static const int wsz[8] = {0, 0, 128 * 9 * 64, 256 * 9 * 128, 256 * 9 * 256, 512 * 9 * 256, 512 * 9 * 512, 512 * 4 * 512};
    for (int i = 2; i <= 7; ++i) { m->wb[i] = a.get<bf16_t>(wsz[i]); m->wtb[i] = a.get<bf16_t>(wsz[i]); }
  }
  {                                                            // bf16 shadows of the recurrent weights (bf16 mode)
    auto sh = [&](ShW& w, size_t R_, size_t C_) {
      w.wb = m->bf16 ? a.get<bf16_t>(R_ * C_) : nullptr; w.wtb = m->bf16 ? a.get<bf16_t>(R_ * C_) : nullptr;
      w.wtf = m->bf16 ? nullptr : a.get<float>(R_ * C_);
    };
    for (int dir = 0; dir < 2; ++dir) for (int l = 0; l < m->Le; ++l) { sh(m->enc[dir][l].swh, 4 * He, He); sh(m->enc[dir][l].swi, 4 * He, l == 0 ? 512 : He); }
    for (int l = 0; l < m->Ld; ++l) { sh(m->dec[l].swi, 4 * Hd, Hd); sh(m->dec[l].swh, 4 * Hd, Hd); }
    sh(m->swa, Hd, Hd); sh(m->swc, Hd, 2 * Hd);
  }
  m->bn_scratch = a.get<char>(bn_scratch_bytes(512)); m->bn_save = a.get<float>(3 * 2 * 512);
  for (int dir = 0; dir < 2; ++dir) {
    for (int l = 0; l < m->Le; ++l) {
      m->ezx[dir][l] = a.get<float>(T * B * 4 * He); m->ehs[dir][l] = a.get<float>((T + 2) * B * He);
      m->ecs[dir][l] = a.get<float>((T + 2) * B * He); m->egates[dir][l] = a.get<float>(T * B * 4 * He);
      m->edz[dir][l] = a.get<float>(T * B * 4 * He);
    }
    for (int l = 0; l < m->Le; ++l) m->edc[dir][l] = a.get<float>(B * He);
    m->edxl[dir] = a.get<float>(T * B * He);
  }
  m->context = a.get<float>(B * T * Hd); m->dctx = a.get<float>(B * T * Hd);
  m->context_b = m->bf16 ? a.get<bf16_t>(B * T * Hd) : nullptr;
  {
    auto hb = [&](size_t n) { return m->bf16 ? a.get<bf16_t>(n) : (bf16_t*)nullptr; };
    m->Xb = hb(T * B * 512);
    for (int dir = 0; dir < 2; ++dir) for (int l = 0; l < m->Le; ++l) { m->ehs_b[dir][l] = hb((T + 2) * B * He); m->edz_b[dir][l] = hb(T * B * 4 * He); }
    for (int l = 0; l < m->Ld; ++l) { m->dhs_b[l] = hb((L + 1) * B * Hd); m->ddz_b[l] = hb(L * B * 4 * Hd); }
    m->out_b = hb((L + 1) * B * Hd); m->cat_b = hb(L * B * 2 * Hd); m->dpre_b = hb(L * B * Hd); m->dq_b = hb(L * B * Hd);
  }
  m->emb_all = a.get<float>(L * B * E); m->zx1_all = a.get<float>(L * B * 4 * Hd); m->emb_seg = a.get<float>((size_t)m->V * 4 * Hd); m->emb_index = a.get<int>(segsum_index_ints((size_t)L * B, m->V));
  for (int l = 0; l < m->Ld; ++l) {
    m->dhs[l] = a.get<float>((L + 1) * B * Hd); m->dcs[l] = a.get<float>((L + 1) * B * Hd);
    m->dgates[l] = a.get<float>(L * B * 4 * Hd); m->ddz[l] = a.get<float>(L * B * 4 * Hd);
    m->dh_rec[l] = a.get<float>(B * Hd); m->dc_st[l] = a.get<float>(B * Hd);
  }
  m->out_all = a.get<float>((L + 1) * B * Hd); m->cat_all = a.get<float>(L * B * 2 * Hd);
  m->q_all = a.get<float>(L * B * Hd); m->a_all = a.get<float>(L * B * T);
  m->logits = a.get<float>(L * B * LOGIT_LD); m->dlogits = a.get<float>(L * B * LOGIT_LD); m->nll_rows = a.get<float>(L * B);
  m->dout_proj = a.get<float>(L * B * Hd); m->dpre_all = a.get<float>(L * B * Hd); m->dcat_all = a.get<float>(L * B * 2 * Hd);
  m->ds_all = a.get<float>(L * B * T); m->dq_all = a.get<float>(L * B * Hd); m->demb_all = a.get<float>(L * B * E);
  m->dfeed = a.get<float>(B * Hd); m->loss_tmp = a.get<float>(64);
  for (int p = 0; p < 2; ++p) {
    for (int l = 0; l < m->Ld; ++l) { m->bc[p][l] = a.get<float>(R * Hd); m->bh[p][l] = a.get<float>(R * Hd); }
    m->bfeed[p] = a.get<float>(R * Hd);
  }
  for (int l = 0; l < m->Ld; ++l) { m->bc_new[l] = a.get<float>(R * Hd); m->bh_new[l] = a.get<float>(R * Hd); }
  if (m->bf16) {
    for (int p = 0; p < 2; ++p) { for (int l = 0; l < m->Ld; ++l) m->bh_b[p][l] = a.get<bf16_t>(R * Hd); m->bfeed_b[p] = a.get<bf16_t>(R * Hd); }
    for (int l = 0; l < m->Ld; ++l) m->bh_new_b[l] = a.get<bf16_t>(R * Hd);
    m->bcat_b = a.get<bf16_t>(R * 2 * Hd);
  }
  m->bzx1 = a.get<float>(R * 4 * Hd); m->bzx_tab = a.get<float>((size_t)m->V * 4 * Hd); m->bq = a.get<float>(R * Hd); m->ba = a.get<float>(R * T);
  m->bcat = a.get<float>(R * 2 * Hd); m->bout = a.get<float>(R * Hd); m->blogits = a.get<float>(R * LOGIT_LD);
  m->blogp = a.get<float>(R * m->V); m->beam_scores = a.get<float>(R);
  m->hist_tok = a.get<int32_t>(L * R); m->hist_par = a.get<int32_t>(L * R);
  m->tgt_pad = a.get<int32_t>(B * L); m->tge_pad = a.get<int32_t>(B * L);
  m->trie_loc[0] = a.get<int32_t>(R); m->trie_loc[1] = a.get<int32_t>(R);
  m->sgd_scratch = a.get<char>(sgd_scratch_bytes());
  // split-K slabs of the filter gradients (bf16 mode): one resident round of workgroups x one fp32 tile each = 64 MiB at most
  m->wg_part_floats = m->bf16 ? (size_t)20 << 20 : 0;      /* 80 MiB: eight k ranges of conv6's 512 x 4608 filter gradient (conv_wgrad_halo_kernel: 32 tiles x 8 = 256 workgroups) */ m->wg_part = m->wg_part_floats ? a.get<float>(m->wg_part_floats) : nullptr;
  if (m->bf16 && He % 64 == 0 && He <= 512) {                    // exchange buffers of the cluster encoder kernels
    // one set of group slots per layer: the layers of a stacked encoder run concurrently (layer wavefront, encoder_forward)
    m->cl_xbytes = m->Le * enc_cluster_xbuf_bytes((int)B, (int)He); m->cl_pbytes = m->Le * enc_cluster_pbuf_bytes((int)B, (int)He);
    m->cl_xbuf = a.get<unsigned long long>(m->cl_xbytes / 8); m->cl_pbuf = a.get<unsigned long long>(m->cl_pbytes / 8);
    m->cl_tbytes = ((size_t)m->Le * 4 * ((B + 15) / 16) * 8 + 64) * 8;    // a layer's 8-row launches start at slot l * 4 * ceil(B / 16) (enc_cluster_forward: gslot * 2) and use 2 * ceil(B / 8) <= 4 * ceil(B / 16) of them
    m->cl_xtab = a.get<unsigned long long>(m->cl_tbytes / 8);      // XCC ids of the members of every group
    m->cl_err = a.get<int>(16 + 256 * 8 + 4096);                 // error flag + the trash slots rows >= B store to + a debugging timeline
    m->bn_snap = a.get<float>(2 * (256 + 512 + 512));            // aocr_bn_state_count() floats
    if (Hd == 1024) m->ctxa_b = a.get<bf16_t>(B * T * Hd);        // the launch chain scores attention against ctx W_a too (round 6: attn_bf16_kernel<..., DUAL>, taken at T <= 64)
    if (Hd == 512 && m->Ld == 2 && m->cfg.input_feed) {          // the decoder loop as one launch (dec_cluster.hip)
      m->dc_xbytes = dec_cluster_xbuf_bytes((int)B); m->dc_tbytes = dec_cluster_xtab_bytes((int)std::max<size_t>(B, 32 * (size_t)dec_chain_beam_group_cap()));      // (beam search on the chain kernel indexes by the group within a launch: up to one group per XCD)
      m->dc_xbuf = a.get<unsigned long long>(m->dc_xbytes / 8); m->dc_xtab = a.get<unsigned long long>(m->dc_tbytes / 8);
      m->ctxa_b = a.get<bf16_t>(B * T * Hd);
      m->dc_bxbytes = dec_cluster_bwd_xbuf_bytes((int)B); m->dc_bxbuf = a.get<unsigned long long>(m->dc_bxbytes / 8);
      m->dc_pbuf = a.get<float>(dec_cluster_pbuf_bytes((int)std::max<size_t>(B, 32 * (size_t)dec_chain_beam_group_cap())) / 4);
    }
  }
  for (int l = 0; l + 1 < m->Ld; ++l) { m->dhm[l] = a.get<float>(L * B * Hd); m->dhm_b[l] = m->bf16 ? a.get<bf16_t>(L * B * Hd) : nullptr; }
  for (int dir = 0; dir < 2; ++dir) for (int l = 0; l + 1 < m->Le; ++l) { m->ehm[dir][l] = a.get<float>(T * B * He); m->ehm_b[dir][l] = m->bf16 ? a.get<bf16_t>(T * B * He) : nullptr; }
  m->shadow_dev = m->bf16 ? a.get<ShadowJob>(128) : nullptr;
  m->ws_bytes = a.off + 256;
  if (base && a.off > bytes) return -1;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// recurrent-step launch helpers: the B operand is a weight matrix W [R][C]; in bf16 mode its bf16 shadow (W for
// y = x W^T, the transposed shadow for y = x W) is read, always K-contiguous; in fp32 mode W itself.
// ------------------------------------------------------------------------------------------------
// bf16 mode: every shadow piece of the model (recurrent matrices; conv weights tap by tap) as one job table
void build_shadow_jobs(aocr_model* m) {
  m->shadow_host.clear(); m->shadow_tiles = 0;
  if (!m->bf16 || !m->shadow_dev) return;
  auto add = [&](const float* w, bf16_t* wb, bf16_t* wtb, int64_t ld, int64_t ldb, int64_t ldt, int R, int C) {
    ShadowJob j{w, wb, wtb, ld, ldb, ldt, R, C, m->shadow_tiles, cdiv(C, 32)};
    m->shadow_tiles += cdiv(C, 32) * cdiv(R, 32);
    m->shadow_host.push_back(j);
  };
  auto up = [&](const ShW& w) { if (w.wb) add(w.w, w.wb, w.wtb, w.ld, w.C, w.R, w.R, w.C); };
  auto conv_jobs = [&](int i) {
    const ConvP& c = m->conv[i]; const int KK = c.ks * c.ks;
    for (int tap = 0; tap < KK; ++tap)
      add(c.w + (size_t)tap * c.cin, m->wb[i] + (size_t)tap * c.cin, m->wtb[i] + (size_t)tap * c.cout, (int64_t)KK * c.cin,
          (int64_t)KK * c.cin, (int64_t)KK * c.cout, c.cout, c.cin);
  };
  conv_jobs(2);                                                  // conv2's taps FIRST: the step launches them on their own (step_prologue) so that conv2 can start behind conv1
  m->shadow_tiles_conv2 = m->shadow_tiles;
  for (int dir = 0; dir < 2; ++dir) for (int l = 0; l < m->Le; ++l) { up(m->enc[dir][l].swh); up(m->enc[dir][l].swi); }
  for (int l = 0; l < m->Ld; ++l) { if (l > 0 || m->cfg.input_feed) up(m->dec[l].swi); up(m->dec[l].swh); }
  up(m->swa); up(m->swc);
  for (int i = 3; i <= 7; ++i) conv_jobs(i);
  if (m->shadow_host.size() > 128) { m->shadow_host.clear(); m->shadow_tiles = 0; m->shadow_tiles_conv2 = 0; }      // table too small: per-matrix launches
}
static void refresh_rnn_shadows(aocr_model* m, hipStream_t st = nullptr) {
  if (!st) st = m->s;
  auto up = [&](const ShW& w) {
    if (w.wb) weight_shadows(st, w.w, w.ld, w.R, w.C, w.wb, w.wtb);
    else if (w.wtf) transpose_f32(st, w.w, w.ld, w.R, w.C, w.wtf);
  };
  for (int dir = 0; dir < 2; ++dir) for (int l = 0; l < m->Le; ++l) { up(m->enc[dir][l].swh); up(m->enc[dir][l].swi); }
  for (int l = 0; l < m->Ld; ++l) { if (l > 0 || m->cfg.input_feed) up(m->dec[l].swi); up(m->dec[l].swh); }
  up(m->swa); up(m->swc);
}
static LoadK bnt_f(const ShW* w0, const ShW* w1) {
  return w1 ? make_loadk2(w0->w, w0->ld, w0->C, w1->w, w1->ld, w1->C, w0->R) : make_loadk(w0->w, w0->ld, w0->R, w0->C);
}
static LoadKh2 bnt_h(const ShW* w0, const ShW* w1) {
  return w1 ? make_loadkh2(w0->wb, w0->C, w0->C, w1->wb, w1->C, w1->C, w0->R) : make_loadkh(w0->wb, w0->C, w0->R, w0->C);
}
static bool hh_ok(const LoadKh2* ah, int nz, int ncols) {        // staged kernel with both operands from bf16 shadows
  if (!ah || ncols % 32 != 0) return false;
  for (int i = 0; i < nz; ++i) if (!ah[i].p0 || ah[i].K <= 0 || ah[i].K % 64 != 0 || ah[i].K0 % 64 != 0) return false;
  return true;
}
// gates forward: z = [x0 ; x1] [W0 ; W1]^T (+ epilogue); nz argument sets; ah = the same A operand as bf16 shadows (optional)
static void run_gates_fwd(aocr_model* m, int nz, const LoadK* a, const ShW* const* w0, const ShW* const* w1, const EpGatesFwd* ep,
                          int M, int H, const LoadKh2* ah = nullptr) {
  if (m->bf16 && hh_ok(ah, nz, H)) {
    // large batch (the reference's default shape): tiled GEMM into a scratch z + elementwise cell pass (ops_gemm.hip: big_step_gates_fwd); scratch = the
    // d z buffer of layer 0, which only the backward pass uses
    if (nz == 1 && big_step_gates_fwd(m->s, ah[0], bnt_h(w0[0], w1[0]), ep[0], M, H, m->ddz[0], (size_t)m->cfg.max_decoder_l * m->cfg.batch_size * 4 * m->Hd)) return;
    GatesFwdArgsHH z[2];
    for (int i = 0; i < nz; ++i) { z[i].a = ah[i]; z[i].b = bnt_h(w0[i], w1[i]); z[i].ep = ep[i]; z[i].K = ah[i].K; }
    launch_small_gates_fwd_hh(m->s, nz, z, M, H);
  } else if (m->bf16) {
    GatesFwdArgsH z[2];
    for (int i = 0; i < nz; ++i) { z[i].a = a[i]; z[i].b = bnt_h(w0[i], w1[i]); z[i].ep = ep[i]; z[i].K = a[i].K; }
    launch_small_gates_fwd_h(m->s, nz, z, M, H);
  } else {
    GatesFwdArgs z[2];
    for (int i = 0; i < nz; ++i) { z[i].a = a[i]; z[i].b = bnt_f(w0[i], w1[i]); z[i].ep = ep[i]; z[i].K = a[i].K; }
    launch_small_gates_fwd(m->s, false, nz, z, M, H);
  }
}
// y = x W^T through an EpStore
static void run_store_nt(aocr_model* m, const LoadK& a, const ShW& w, const EpStore& ep, int M, const LoadKh2* ah = nullptr) {
  if (m->bf16 && hh_ok(ah, 1, w.R)) { if (big_step_store(m->s, *ah, bnt_h(&w, nullptr), ep, M, w.R)) return; SmallArgsHH z; z.a = *ah; z.b = bnt_h(&w, nullptr); z.ep = ep; z.K = ah->K; launch_small_hh(m->s, 1, &z, M, w.R); }
  else if (m->bf16) { SmallArgsH z; z.a = a; z.b = bnt_h(&w, nullptr); z.ep = ep; z.K = a.K; launch_small_h(m->s, 1, &z, M, w.R); }
  else { SmallKKArgs z; z.a = a; z.b = bnt_f(&w, nullptr); z.ep = ep; z.K = a.K; launch_small_kk(m->s, false, 1, &z, M, w.R); }
}
// y = x W through an EpStore (x is [M][R], y is [M][C])
static void run_store_nn(aocr_model* m, const LoadK& a, const ShW& w, const EpStore& ep, int M, const LoadKh2* ah = nullptr) {
  if (m->bf16 && hh_ok(ah, 1, w.C)) { if (big_step_store(m->s, *ah, make_loadkh(w.wtb, w.R, w.C, w.R), ep, M, w.C)) return; SmallArgsHH z; z.a = *ah; z.b = make_loadkh(w.wtb, w.R, w.C, w.R); z.ep = ep; z.K = ah->K; launch_small_hh(m->s, 1, &z, M, w.C); }
  else if (m->bf16) { SmallArgsH z; z.a = a; z.b = make_loadkh(w.wtb, w.R, w.C, w.R); z.ep = ep; z.K = a.K; launch_small_h(m->s, 1, &z, M, w.C); }
  else if (w.wtf) { SmallKKArgs z; z.a = a; z.b = make_loadk(w.wtf, w.R, w.C, w.R); z.ep = ep; z.K = a.K; launch_small_kk(m->s, false, 1, &z, M, w.C); }
  else { SmallKMNArgs z; z.a = a; z.b = make_loadmn(w.w, w.ld, w.C, a.K); z.ep = ep; z.K = a.K; launch_small_kmn(m->s, false, 1, &z, M, w.C); }
}
// up to three independent y_i = x_i W_i of the same shape in ONE launch (bf16 shadows on both sides); otherwise one launch each
static void run_store_nn_group(aocr_model* m, int n, const LoadK* a, const ShW* const* w, const EpStore* ep, int M, const LoadKh2* ah) {
  bool same = m->bf16 && hh_ok(ah, n, w[0]->C);
  for (int i = 1; i < n && same; ++i) same = w[i]->C == w[0]->C;
  if (same) {
    SmallArgsHH z[3];
    for (int i = 0; i < n; ++i) { z[i].a = ah[i]; z[i].b = make_loadkh(w[i]->wtb, w[i]->R, w[i]->C, w[i]->R); z[i].ep = ep[i]; z[i].K = ah[i].K; }
    launch_small_hh(m->s, n, z, M, w[0]->C);
    return;
  }
  bool same32 = !m->bf16;                               // fp32 mode: the same grouping over the fp32 transposed weights
  for (int i = 0; i < n && same32; ++i) same32 = w[i]->wtf != nullptr && w[i]->C == w[0]->C;
  if (same32) {
    SmallKKArgs z[3];
    for (int i = 0; i < n; ++i) { z[i].a = a[i]; z[i].b = make_loadk(w[i]->wtf, w[i]->R, w[i]->C, w[i]->R); z[i].ep = ep[i]; z[i].K = a[i].K; }
    launch_small_kk(m->s, false, n, z, M, w[0]->C);
    return;
  }
  for (int i = 0; i < n; ++i) run_store_nn(m, a[i], *w[i], ep[i], M, ah ? &ah[i] : nullptr);
}
// gate backward: d(h) GEMM part = x W (K may be 0: no GEMM part)
static void run_gates_bwd(aocr_model* m, int nz, const LoadK* a, const ShW* const* w, const EpGatesBwd* ep, int M, int H,
                          const LoadKh2* ah = nullptr) {
  if (nz == 1 && a[0].K == 0 && ep[0].drop.thr == 0) { gates_elem_bwd(m->s, ep[0], M, H); return; }      // no GEMM part (round 6: the chain's top cell when the attention backward kernel supplies d h's attention part)
  if (m->bf16 && hh_ok(ah, nz, H)) {
    GatesBwdArgsHH z[2];
    for (int i = 0; i < nz; ++i) { z[i].a = ah[i]; z[i].b = make_loadkh(w[i]->wtb, w[i]->R, w[i]->C, ah[i].K); z[i].ep = ep[i]; z[i].K = ah[i].K; }
    launch_small_gates_bwd_hh(m->s, nz, z, M, H);
  } else if (m->bf16) {
    GatesBwdArgsH z[2];
    for (int i = 0; i < nz; ++i) { z[i].a = a[i]; z[i].b = make_loadkh(w[i]->wtb, w[i]->R, w[i]->C, a[i].K); z[i].ep = ep[i]; z[i].K = a[i].K; }
    launch_small_gates_bwd_h(m->s, nz, z, M, H);
  } else {
    if (w[0]->wtf) {                                          // fp32 mode with K-contiguous transposed weights
      GatesBwdKKArgs z[2];
      for (int i = 0; i < nz; ++i) { z[i].a = a[i]; z[i].b = make_loadk(w[i]->wtf, w[i]->R, w[i]->C, a[i].K); z[i].ep = ep[i]; z[i].K = a[i].K; }
      launch_small_gates_bwd_kk(m->s, nz, z, M, H);
      return;
    }
    GatesBwdArgs z[2];
    for (int i = 0; i < nz; ++i) { z[i].a = a[i]; z[i].b = make_loadmn(w[i]->w, w[i]->ld, w[i]->C, a[i].K); z[i].ep = ep[i]; z[i].K = a[i].K; }
    launch_small_gates_bwd(m->s, false, nz, z, M, H);
  }
}

// ------------------------------------------------------------------------------------------------
// CNN forward, cnn.lua:9-45.  Output X is time-major (T,B,512) (= cnn_output:transpose(1,2), model.lua:288).
// ------------------------------------------------------------------------------------------------
static int bn_sync_allreduce(void* ctx, void* buf, int64_t count, int dtype, hipStream_t s) { return comm_allreduce((aocr_model*)ctx, buf, count, dtype, s, 1); }

void cnn_forward(aocr_model* m, const float* images, const Dims& d, int training, int update_running) {
  hipStream_t s = m->s; const bool bf = m->bf16; const int B = d.B;
  m->y16[0] = m->y16[1] = 0;                                     // only a training conv_forward below sets them again: the evaluation / folded branch leaves Y3 / Y5 untouched, never "bf16 from an older step"
  const BnSync bsync_v{bn_sync_allreduce, m}; const BnSync* bsync = (training && sync_bn_on(m)) ? &bsync_v : nullptr;
  if (bf && !m->shadow_host.empty()) {                          // every bf16 shadow of the step in one launch
    if (!m->shadow_pending) shadow_jobs(s, m->shadow_dev, (int)m->shadow_host.size(), m->shadow_tiles);      // (pending: step_prologue put it on the side stream)
  } else {
    refresh_rnn_shadows(m);
    for (int i = 2; i <= 7; ++i) {                              // refresh the re-laid weight copies (weights change every step)
      if (bf) conv_weight_shadows(s, m->conv[i].w, m->wb[i], m->wtb[i], m->conv[i].cout, m->conv[i].ks * m->conv[i].ks, m->conv[i].cin);
      else conv_weight_transpose_f32(s, m->conv[i].w, m->wtf[i], m->conv[i].cout, m->conv[i].ks * m->conv[i].ks, m->conv[i].cin);
    }
  }
  // bf16 mode: the pooled conv outputs exist only as bf16 shadows (every consumer -- next conv, filter gradient, ReLU mask of the
  // un-pool -- reads the shadow; aocr_get_tensor materialises fp32 on demand)
  prof_mark(m, AOCR_PROF_POOL_CONV1); conv1_forward(s, images, m->conv[1].w, m->conv[1].b, bf ? nullptr : m->A1, B, d.H, d.W, m->A1b, training ? m->route1 : nullptr);
  m->route1_valid = training;                                   // conv1_backward takes the pooling/ReLU decisions from here instead of re-evaluating the layer
  if (m->shadow_pending && m->shadow2_pending) { hipStreamWaitEvent(s, m->shadow2_done, 0); m->shadow2_pending = false; }       // conv1 reads no shadow; conv2 reads its own taps (the first part of the table)
  else if (m->shadow_pending) { hipStreamWaitEvent(s, m->shadow_done, 0); m->shadow_pending = false; }
  prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A1, m->conv[2].w, m->conv[2].b, bf ? nullptr : m->A2, m->idx2, B, d.H1, d.W1, 64, 128, 3, 1, 1, 1, m->A1b, m->wb[2], m->A2b);
  if (m->shadow_pending) { hipStreamWaitEvent(s, m->shadow_done, 0); m->shadow_pending = false; }       // the rest of the table (conv3 .. conv7, the recurrent matrices): refreshed under conv2
  // evaluation mode, bf16: BatchNorm + ReLU of conv3 / conv5 are a per-channel affine map -> folded into the conv epilogue, which then writes only
  // the bf16 shadow (no fp32 map, no apply pass); conv7's BatchNorm also transposes to (T, B) and stays a pass of its own
  const bool fold = bf && !training && !getenv("AOCR_NO_BN_FOLD");
  if (fold) {
    prof_mark(m, AOCR_PROF_BN); bn_eval_prepare(s, m->bn[3].rm, m->bn[3].rv, m->bn[3].save, 256);
    prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A2, m->conv[3].w, m->conv[3].b, nullptr, nullptr, B, d.H2, d.W2, 128, 256, 3, 1, 0, 0, m->A2b, m->wb[3], m->A3b, 0,
                                                   m->bn[3].save, m->bn[3].w, m->bn[3].b);
  } else {
  int bnc = 0;                                              // > 0: the conv's staged epilogue left the BatchNorm's partial sums in bn_scratch (training only)
  prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A2, m->conv[3].w, m->conv[3].b, m->Y3, nullptr, B, d.H2, d.W2, 128, 256, 3, 1, 0, 0, m->A2b, m->wb[3], nullptr, 0,
                                                 nullptr, nullptr, nullptr, training ? (double*)m->bn_scratch : nullptr, &bnc, &m->y16[0]);
  prof_mark(m, AOCR_PROF_BN); bn_relu_forward(s, m->Y3, bf ? nullptr : m->A3, m->bn[3].w, m->bn[3].b, m->bn[3].rm, m->bn[3].rv, m->bn[3].save, m->bn_scratch,
                  (int64_t)B * d.H2 * d.W2, 256, training, update_running, 0, m->A3b, bsync, bnc, m->y16[0] ? reinterpret_cast<const bf16_t*>(m->Y3) : nullptr);
  }
  prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A3, m->conv[4].w, m->conv[4].b, bf ? nullptr : m->A4, m->idx4, B, d.H2, d.W2, 256, 256, 3, 1, 1, 2, m->A3b, m->wb[4], m->A4b);
  if (fold) {
    prof_mark(m, AOCR_PROF_BN); bn_eval_prepare(s, m->bn[5].rm, m->bn[5].rv, m->bn[5].save, 512);
    prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A4, m->conv[5].w, m->conv[5].b, nullptr, nullptr, B, d.H4, d.W2, 256, 512, 3, 1, 0, 0, m->A4b, m->wb[5], m->A5b, 0,
                                                   m->bn[5].save, m->bn[5].w, m->bn[5].b);
  } else {
  int bnc = 0;
  prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A4, m->conv[5].w, m->conv[5].b, m->Y5, nullptr, B, d.H4, d.W2, 256, 512, 3, 1, 0, 0, m->A4b, m->wb[5], nullptr, 0,
                                                 nullptr, nullptr, nullptr, training ? (double*)m->bn_scratch : nullptr, &bnc, &m->y16[1]);
  prof_mark(m, AOCR_PROF_BN); bn_relu_forward(s, m->Y5, bf ? nullptr : m->A5, m->bn[5].w, m->bn[5].b, m->bn[5].rm, m->bn[5].rv, m->bn[5].save, m->bn_scratch,
                  (int64_t)B * d.H4 * d.W2, 512, training, update_running, 0, m->A5b, bsync, bnc, m->y16[1] ? reinterpret_cast<const bf16_t*>(m->Y5) : nullptr);
  }
  prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A5, m->conv[6].w, m->conv[6].b, bf ? nullptr : m->A6, m->idx6, B, d.H4, d.W2, 512, 512, 3, 1, 1, 2, m->A5b, m->wb[6], m->A6b);
  prof_mark(m, AOCR_PROF_CONV_FWD); conv_forward(s, bf, m->A6, m->conv[7].w, m->conv[7].b, m->Y7, nullptr, B, d.H6, d.W2, 512, 512, 2, 0, 0, 0, m->A6b, m->wb[7], nullptr);
  prof_mark(m, AOCR_PROF_BN); bn_relu_forward(s, m->Y7, m->X, m->bn[7].w, m->bn[7].b, m->bn[7].rm, m->bn[7].rv, m->bn[7].save, m->bn_scratch,
                  (int64_t)B * d.T, 512, training, update_running, B, m->Xb, bsync);
}

// the filter gradients of the CNN backward pass run on the side stream (cnn_backward; backward_all asks too: the hoisted recurrent weight gradients then need no
// join in front of the CNN backward pass -- the filter gradients queue behind them on the same stream)
static bool cnn_wgrad_on_side(aocr_model* m) {
  auto evok = [](hipEvent_t& e) { return e || hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
  return m->bf16 && m->G2b && !getenv("AOCR_DBG_STOP") && !m->prof_on && m->side && m->side_done && !getenv("AOCR_NO_SIDE_WGRAD") && !env_on("AOCR_NO_CNN_WGRAD_SIDE") &&
         evok(m->cw_map[0]) && evok(m->cw_map[1]) && evok(m->cw_done[0]) && evok(m->cw_done[1]) && evok(m->cw_main);
}

static void cnn_backward(aocr_model* m, const float* images, const Dims& d) {
  hipStream_t s = m->s; const bool bf = m->bf16; const int B = d.B;
  const BnSync bsync_v{bn_sync_allreduce, m}; const BnSync* bsync = sync_bn_on(m) ? &bsync_v : nullptr;
  float *G0 = m->G0, *G1 = m->G1;
  // Round 4: the filter gradients on the side stream.  A stage is E_k (BatchNorm backward / un-pool: HBM-bound, writes the gradient map d Y_k) ->
  // { filter gradient k, data gradient k } (both MFMA-bound, both read d Y_k) -> E_(k-1).  On one stream the elementwise passes (0.5 ms of the
  // backward pass at C3) run with the MFMA pipes idle; with the filter gradient of stage k on the (low-priority) side stream it runs beside the data
  // gradient k and E_(k-1).  d Y alternates between two bf16 buffers so that E_(k-1) can write the next map while the filter gradient k still reads
  // its own; an event per buffer keeps E_(k-2) from overwriting it before that filter gradient is done.  AOCR_NO_CNN_WGRAD_SIDE=1: one stream.
  bf16_t* Gb[2] = {m->G0b, m->G2b}; int gp = 0;
  const char* dbg_stop = getenv("AOCR_DBG_STOP");          // debugging aid: leave the gradient map of a stage in place (tap "g0")
  const int stop = dbg_stop ? atoi(dbg_stop) : 0;
  const bool ws = !stop && cnn_wgrad_on_side(m);
  hipStream_t sw = ws ? m->side : s;
  const bool wg_after = ws && env_on("AOCR_CNN_WGRAD_AFTER_DGRAD");      // A/B: start the filter gradient of a stage behind its data gradient (beside the next elementwise pass only)
  if (!ws) Gb[1] = Gb[0];
  auto map_ready = [&]() { if (ws) { hipEventRecord(m->cw_map[gp], s); hipStreamWaitEvent(sw, m->cw_map[gp], 0); } };        // d Y_k (Gb[gp]) is complete: the side stream may read it
  auto wgrad_done = [&]() { if (ws) hipEventRecord(m->cw_done[gp], sw); };
  auto next_map = [&]() { if (ws) { gp ^= 1; hipStreamWaitEvent(s, m->cw_done[gp], 0); } };                                    // the next E writes Gb[gp]: the filter gradient that read it two stages ago is done
  int bnbc = 0;                                                  // > 0: the data gradient's epilogue left the BatchNorm backward's partial sums in bn_scratch (conv_backward_data: bnb_chunks)
  int g16 = 0; const bf16_t* const G1h = reinterpret_cast<const bf16_t*>(G1);      // conv_backward_data left G1 as bf16 (its dx16): the BatchNorm backward / un-pool pass that follows reads it so
  // bf16 mode: fp32 G0 is free for the whole pass (every gradient map lives in its bf16 shadow), so the eight partial slabs of the fused bias /
  // conv1 gradients get regions of their own and their column sums are finished by TWO launches (before the bucket event, at the end)
  // instead of eight dependent ~10 us dispatches
  constexpr size_t SLAB = (size_t)4 << 20;                      // floats per slab: >= 2048 x 1024 (BatchNorm), 4096 x 640 (conv1)
  ColsumJobs cj; cj.n = 0; cj.total = 0;
  ColsumJobs* defer = (bf && m->gmax >= 8 * SLAB && !getenv("AOCR_NO_COLSUM_DEFER")) ? &cj : nullptr;
  auto slab = [&](int k) { return defer ? G0 + (size_t)k * SLAB : G0; };
  // bn7 + relu (dX is time-major, Y7 batch-major)
  // bf16 mode: the BatchNorm backward writes only the bf16 shadow of its gradient, takes the ReLU mask from the bf16 output
  // shadow and accumulates the preceding conv's bias gradient (fp32 G0 is free there: partial slab)
  prof_mark(m, AOCR_PROF_BN); bn_relu_backward(s, m->Y7, m->X, m->dX, m->bn[7].w, m->bn[7].save, bf ? nullptr : G0, m->bn[7].dw, m->bn[7].db, m->bn_scratch,
                   (int64_t)B * d.T, 512, B, Gb[gp], bf ? m->Xb : nullptr, bf ? m->conv[7].db : nullptr, bf ? slab(0) : nullptr, bsync, defer);
  if (stop == 1) { if (defer) colsum_flush(s, cj); return; }
  { auto W = [&]() { prof_mark(m, AOCR_PROF_CONV_WGRAD); conv_backward_filter(sw, bf, m->A6, G0, m->conv[7].dw, bf ? nullptr : m->conv[7].db, B, d.H6, d.W2, 512, 512, 2, 0, m->A6b, Gb[gp], m->wg_part, m->wg_part_floats); };
    auto D = [&]() { prof_mark(m, AOCR_PROF_CONV_DGRAD); conv_backward_data(s, bf, G0, m->conv[7].w, G1, B, d.H6, d.W2, 512, 512, 2, 0, Gb[gp], m->wtb[7], m->wtf[7], bf ? &g16 : nullptr); };
    if (wg_after) { D(); map_ready(); W(); wgrad_done(); } else { map_ready(); W(); wgrad_done(); D(); } }
  next_map(); prof_mark(m, AOCR_PROF_POOL_CONV1); unpool_relu_backward(s, G1, m->A6, m->idx6, bf ? nullptr : G0, B, d.H4, d.W2, 512, 2, Gb[gp], bf ? m->conv[6].db : nullptr, bf ? slab(1) : nullptr, bf ? m->A6b : nullptr, defer, g16 ? G1h : nullptr);   // bf16: shadow only + fused bias gradient (fp32 G0 is free: partial slab)
  if (stop == 2) { if (defer) colsum_flush(s, cj); return; }
  { auto W = [&]() { prof_mark(m, AOCR_PROF_CONV_WGRAD); conv_backward_filter(sw, bf, m->A5, G0, m->conv[6].dw, bf ? nullptr : m->conv[6].db, B, d.H4, d.W2, 512, 512, 3, 1, m->A5b, Gb[gp], m->wg_part, m->wg_part_floats); };
    auto D = [&]() { prof_mark(m, AOCR_PROF_CONV_DGRAD); { const BnBwdFuse bf5{m->Y5, m->A5b, m->bn[5].save, (double*)m->bn_scratch}; conv_backward_data(s, bf, G0, m->conv[6].w, G1, B, d.H4, d.W2, 512, 512, 3, 1, Gb[gp], m->wtb[6], m->wtf[6], bf ? &g16 : nullptr, (bf && !m->y16[1]) ? &bf5 : nullptr, &bnbc); } };
    if (wg_after) { D(); map_ready(); W(); wgrad_done(); } else { map_ready(); W(); wgrad_done(); D(); } }
  next_map(); prof_mark(m, AOCR_PROF_BN); bn_relu_backward(s, m->Y5, m->A5, G1, m->bn[5].w, m->bn[5].save, bf ? nullptr : G0, m->bn[5].dw, m->bn[5].db, m->bn_scratch,
                   (int64_t)B * d.H4 * d.W2, 512, 0, Gb[gp], bf ? m->A5b : nullptr, bf ? m->conv[5].db : nullptr, bf ? slab(2) : nullptr, bsync, defer, m->y16[1] ? reinterpret_cast<const bf16_t*>(m->Y5) : nullptr, g16 ? G1h : nullptr, bnbc); bnbc = 0;
  { auto W = [&]() { prof_mark(m, AOCR_PROF_CONV_WGRAD); conv_backward_filter(sw, bf, m->A4, G0, m->conv[5].dw, bf ? nullptr : m->conv[5].db, B, d.H4, d.W2, 256, 512, 3, 1, m->A4b, Gb[gp], m->wg_part, m->wg_part_floats); };
    auto D = [&]() { prof_mark(m, AOCR_PROF_CONV_DGRAD); conv_backward_data(s, bf, G0, m->conv[5].w, G1, B, d.H4, d.W2, 256, 512, 3, 1, Gb[gp], m->wtb[5], m->wtf[5], bf ? &g16 : nullptr); };
    if (wg_after) { D(); map_ready(); W(); wgrad_done(); } else { map_ready(); W(); wgrad_done(); }
  if (defer) colsum_flush(s, cj);                               // conv7.b, conv6.b, conv5.b
  if (ws) { hipEventRecord(m->cw_main, s); hipStreamWaitEvent(sw, m->cw_main, 0); hipEventRecord(m->grad_ev[2], sw); }      // (behind the filter gradient of conv5 on the side stream AND the bias sums on this one)
  else hipEventRecord(m->grad_ev[2], s);                        // every CNN gradient from conv5.w upwards is complete
    if (!wg_after) D(); }
  next_map(); prof_mark(m, AOCR_PROF_POOL_CONV1); unpool_relu_backward(s, G1, m->A4, m->idx4, bf ? nullptr : G0, B, d.H2, d.W2, 256, 2, Gb[gp], bf ? m->conv[4].db : nullptr, bf ? slab(3) : nullptr, bf ? m->A4b : nullptr, defer, g16 ? G1h : nullptr);   // bf16: shadow only + fused bias gradient (fp32 G0 is free: partial slab)
  { auto W = [&]() { prof_mark(m, AOCR_PROF_CONV_WGRAD); conv_backward_filter(sw, bf, m->A3, G0, m->conv[4].dw, bf ? nullptr : m->conv[4].db, B, d.H2, d.W2, 256, 256, 3, 1, m->A3b, Gb[gp], m->wg_part, m->wg_part_floats); };
    auto D = [&]() { prof_mark(m, AOCR_PROF_CONV_DGRAD); { const BnBwdFuse bf3{m->Y3, m->A3b, m->bn[3].save, (double*)m->bn_scratch}; conv_backward_data(s, bf, G0, m->conv[4].w, G1, B, d.H2, d.W2, 256, 256, 3, 1, Gb[gp], m->wtb[4], m->wtf[4], bf ? &g16 : nullptr, (bf && !m->y16[0]) ? &bf3 : nullptr, &bnbc); } };
    if (wg_after) { D(); map_ready(); W(); wgrad_done(); } else { map_ready(); W(); wgrad_done(); D(); } }
  next_map(); prof_mark(m, AOCR_PROF_BN); bn_relu_backward(s, m->Y3, m->A3, G1, m->bn[3].w, m->bn[3].save, bf ? nullptr : G0, m->bn[3].dw, m->bn[3].db, m->bn_scratch,
                   (int64_t)B * d.H2 * d.W2, 256, 0, Gb[gp], bf ? m->A3b : nullptr, bf ? m->conv[3].db : nullptr, bf ? slab(4) : nullptr, bsync, defer, m->y16[0] ? reinterpret_cast<const bf16_t*>(m->Y3) : nullptr, g16 ? G1h : nullptr, bnbc); bnbc = 0;
  { auto W = [&]() { prof_mark(m, AOCR_PROF_CONV_WGRAD); conv_backward_filter(sw, bf, m->A2, G0, m->conv[3].dw, bf ? nullptr : m->conv[3].db, B, d.H2, d.W2, 128, 256, 3, 1, m->A2b, Gb[gp], m->wg_part, m->wg_part_floats); };
    auto D = [&]() { prof_mark(m, AOCR_PROF_CONV_DGRAD); conv_backward_data(s, bf, G0, m->conv[3].w, G1, B, d.H2, d.W2, 128, 256, 3, 1, Gb[gp], m->wtb[3], m->wtf[3]); };
    if (wg_after) { D(); map_ready(); W(); wgrad_done(); } else { map_ready(); W(); wgrad_done(); D(); } }
  next_map(); prof_mark(m, AOCR_PROF_POOL_CONV1); unpool_relu_backward(s, G1, m->A2, m->idx2, bf ? nullptr : G0, B, d.H1, d.W1, 128, 1, Gb[gp], bf ? m->conv[2].db : nullptr, bf ? slab(5) : nullptr, bf ? m->A2b : nullptr, defer);   // bf16: shadow only + fused bias gradient (fp32 G0 is free: partial slab)
  { auto W = [&]() { prof_mark(m, AOCR_PROF_CONV_WGRAD); conv_backward_filter(sw, bf, m->A1, G0, m->conv[2].dw, bf ? nullptr : m->conv[2].db, B, d.H1, d.W1, 64, 128, 3, 1, m->A1b, Gb[gp], m->wg_part, m->wg_part_floats); };
    auto D = [&]() { prof_mark(m, AOCR_PROF_CONV_DGRAD); conv_backward_data(s, bf, G0, m->conv[2].w, G1, B, d.H1, d.W1, 64, 128, 3, 1, Gb[gp], m->wtb[2], m->wtf[2]); };
    if (wg_after) { D(); map_ready(); W(); wgrad_done(); } else { map_ready(); W(); wgrad_done(); D(); } }
  prof_mark(m, AOCR_PROF_POOL_CONV1); conv1_backward(s, images, m->conv[1].w, m->conv[1].b, G1, m->conv[1].dw, m->conv[1].db, B, d.H, d.W,
                 (size_t)B * d.H1 * d.W1 * 128 >= (size_t)4096 * 640 ? slab(6) : nullptr, defer,      // G0 is free here: use it as the partial slab
                 m->route1_valid && !getenv("AOCR_CONV1_RECOMPUTE") ? m->route1 : nullptr);
  if (defer) colsum_flush(s, cj);                               // conv4.b, conv3.b, conv2.b, conv1.w, conv1.b
  if (ws) { hipEventRecord(m->side_done, sw); hipStreamWaitEvent(s, m->side_done, 0); }
}

// ------------------------------------------------------------------------------------------------
// encoder, model.lua:291-316.  State slots: index t+1 holds step t; slot 0 / slot T+1 are the zero initial
// states of the forward / backward direction.
// ------------------------------------------------------------------------------------------------
// The whole-sequence encoder kernels (rnn_seq.hip) need bf16 shadows, B % 16 == 0, He in {64,128,256} and one CU per
// workgroup; AOCR_NO_SEQ=1 forces the per-step kernels (parity tests compare the two).
static bool seq_kernels_ok(const aocr_model* m, int B) {
  if (!m->bf16 || !m->ehs_b[0][0] || !m->enc[0][0].swh.wb) return false;
  const char* e = getenv("AOCR_NO_SEQ");
  if (e && e[0] == '1') return false;
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return enc_seq_supported(B, m->He, cus);
}

// The cluster kernels (rnn_cluster.hip) need bf16 shadows and He in {64,128,256,512}; any batch size.  AOCR_NO_CLUSTER=1 falls back to
// the one-workgroup-per-16-rows kernels / the per-step kernels (parity tests compare the paths).
static bool cluster_ok(const aocr_model* m, int B, int T, int& G, int& RT, int& groups) {
  if (!m->bf16 || !m->cl_xbuf || !m->ehs_b[0][0] || !m->enc[0][0].swh.wb || !m->edz_b[0][0]) return false;
  const char* e = getenv("AOCR_NO_CLUSTER");
  if (e && e[0] == '1') return false;
  e = getenv("AOCR_NO_SEQ");                              // "no whole-sequence kernels at all": the per-step launch chain
  if (e && e[0] == '1') return false;
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return enc_cluster_plan(B, m->He, T, cus - comm_reserved_cus(m), G, RT, groups);
}
static unsigned next_epoch(aocr_model* m) {
  if (++m->cl_epoch >= (1u << 20)) {                              // tags would repeat: clear the buffers and start over
    hipMemsetAsync(m->cl_xbuf, 0, m->cl_xbytes, m->s); hipMemsetAsync(m->cl_pbuf, 0, m->cl_pbytes, m->s); m->cl_epoch = 1;
    hipMemsetAsync(m->cl_xtab, 0, m->cl_tbytes, m->s);
    if (m->dc_xbuf) { hipMemsetAsync(m->dc_xbuf, 0, m->dc_xbytes, m->s); hipMemsetAsync(m->dc_xtab, 0, m->dc_tbytes, m->s); hipMemsetAsync(m->dc_bxbuf, 0, m->dc_bxbytes, m->s); }
  }
  return m->cl_epoch;
}

// Dropout site codes (stream = 64 * train step + site): decoder layer L's input 2..4, attention output 16, encoder fw / bw layer L's input 32 + L / 48 + L
static DropSpec drop_site(const aocr_model* m, int site, long long off) {
  DropSpec d;
  if (!m->drop_on) return d;
  const unsigned long long stream = m->drop_step * 64ull + (unsigned long long)site;
  d.base = splitmix64_(m->drop_seed ^ (stream * 0xD1342543DE82EF95ull)); d.thr = m->drop_thr; d.scale = (float)(1.0 / (1.0 - m->drop_p)); d.off = off;
  return d;
}

// ---- layer wavefront of a stacked encoder (Le >= 2) ----------------------------------------------------------------------------
// The fw / bw stacks are independent (model.lua:291-316: layer l of a direction reads only layer l-1 of the SAME direction), and a
// cluster launch keeps 2 x groups x He/64 compute units busy (C5: 16 of 256) for T latency-bound steps.  So the sequence is cut into
// C chunks of iterations and layer l runs chunk c (its hoisted input GEMM over the chunk's rows, then the cluster kernel with
// it0 / it1) on its own stream as soon as layer l-1 has finished chunk c: Le layers take (C + Le - 1) / C sequence times instead of Le.
// State crosses the chunk boundary through the slots the kernels write anyway (c, bf16 h; d c, bf16 d z on the way back).
// Off under the family profile (prof_on: one stream), AOCR_NO_LAYER_PIPE=1, or when the layers' groups do not fit the chip at once.
// AOCR_LAYER_PIPE_CHUNKS=n forces the chunk count (tests: short sequences).
static int layer_pipe_chunks(const aocr_model* m, int T, int G, int groups) {
  if (m->Le < 2 || m->prof_on || getenv("AOCR_NO_LAYER_PIPE")) return 0;
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  if (2 * groups * G * m->Le > cus) return 0;
  int C = std::min(8, T / 48);
  if (const char* e = getenv("AOCR_LAYER_PIPE_CHUNKS")) C = std::min(atoi(e), T);
  return C >= 2 ? C : 0;
}
static bool layer_pipe_streams(aocr_model* m, int n_events) {
  for (int l = 1; l < m->Le; ++l)
    if (!m->lay_s[l] && hipStreamCreateWithFlags(&m->lay_s[l], hipStreamNonBlocking) != hipSuccess) { m->lay_s[l] = nullptr; return false; }
  while ((int)m->lay_ev.size() < n_events) {
    hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    m->lay_ev.push_back(e);
  }
  return true;
}
// the exchange buffers are cleared on the main stream when the launch epochs wrap: never while another stream is inside a cluster kernel
static void reserve_epochs(aocr_model* m, unsigned n) { if (m->cl_epoch + n + 1 >= (1u << 20)) { m->cl_epoch = (1u << 20); (void)next_epoch(m); } }

static void encoder_forward_pipe(aocr_model* m, const Dims& d, int C, int clG, int clRT, int clGroups) {
  hipStream_t s0 = m->s;
  const int B = d.B, T = d.T, He = m->He, Hd = m->Hd, Le = m->Le;
  const size_t slot = (size_t)B * He;
  for (int l = 0; l < Le; ++l) {                          // zero initial states of every layer; layer 0's input part for the whole sequence
    ZeroList zl;
    for (int dir = 0; dir < 2; ++dir) {
      zl.add(m->ehs[dir][l], slot * sizeof(float)); zl.add(m->ehs[dir][l] + (size_t)(T + 1) * slot, slot * sizeof(float));
      zl.add(m->ecs[dir][l], slot * sizeof(float)); zl.add(m->ecs[dir][l] + (size_t)(T + 1) * slot, slot * sizeof(float));
      zl.add(m->ehs_b[dir][l], slot * sizeof(bf16_t)); zl.add(m->ehs_b[dir][l] + (size_t)(T + 1) * slot, slot * sizeof(bf16_t));
    }
    zero_many(s0, zl);
  }
  for (int dir = 0; dir < 2; ++dir) {
    const LstmP& p = m->enc[dir][0];
    gemm_hh(s0, m->Xb, p.in, p.swi.wb, p.in, m->ezx[dir][0], 4 * He, T * B, 4 * He, p.in, p.bi, p.bh, 0);
  }
  reserve_epochs(m, (unsigned)(C * Le));
  hipEvent_t* ev = m->lay_ev.data();                       // [l * C + c]: layer l finished chunk c; [Le * C + l]: joins
  hipEventRecord(ev[Le * C], s0);
  for (int l = 1; l < Le; ++l) hipStreamWaitEvent(m->lay_s[l], ev[Le * C], 0);
  const int Tc = (T + C - 1) / C;
  for (int c = 0; c < C; ++c) {
    const int it0 = c * Tc, it1 = std::min(T, it0 + Tc);
    if (it0 >= it1) break;
    for (int l = 0; l < Le; ++l) {
      hipStream_t sl = l == 0 ? s0 : m->lay_s[l];
      if (l > 0) {
        hipStreamWaitEvent(sl, ev[(l - 1) * C + c], 0);
        for (int dir = 0; dir < 2; ++dir) {                // the input part of this chunk's steps: t = it (fw) / T-1-it (bw)
          const LstmP& p = m->enc[dir][l];
          const int t_lo = dir ? T - it1 : it0, n = it1 - it0;
          const bf16_t* xinb = m->ehs_b[dir][l - 1] + (size_t)(t_lo + 1) * slot;
          if (m->drop_on) {                                // LSTM.lua:68-69: x = Dropout(h of the layer below)
            dropout_apply_b(sl, xinb, m->ehm[dir][l - 1] + (size_t)t_lo * slot, m->ehm_b[dir][l - 1] + (size_t)t_lo * slot, (int64_t)n * slot,
                            drop_site(m, (dir ? 48 : 32) + l + 1, (long long)t_lo * (long long)slot));
            xinb = m->ehm_b[dir][l - 1] + (size_t)t_lo * slot;
          }
          gemm_hh(sl, xinb, p.in, p.swi.wb, p.in, m->ezx[dir][l] + (size_t)t_lo * B * 4 * He, 4 * He, n * B, 4 * He, p.in, p.bi, p.bh, 0);
        }
      }
      const bool top = l == Le - 1;
      EncClFwdArgs a; a.B = B; a.T = T; a.He = He; a.Hd = Hd; a.groups = clGroups; a.epoch = next_epoch(m); a.xbuf = m->cl_xbuf; a.err = m->cl_err; a.xtab = m->cl_xtab;
      a.it0 = it0; a.it1 = it1; a.gslot = l * 2 * clGroups;
      for (int dir = 0; dir < 2; ++dir) {
        EncSeqDir& e = a.d[dir];
        e.w = m->enc[dir][l].swh.wb; e.zx = m->ezx[dir][l]; e.hs = m->ehs[dir][l]; e.cs = m->ecs[dir][l]; e.hsb = m->ehs_b[dir][l];
        e.gates = m->egates[dir][l]; e.ctx = top ? m->context + dir * He : nullptr; e.reverse = dir;
      }
      enc_cluster_forward(sl, a, clG, clRT, comm_reserved_cus(m), Le);
      if (!top) hipEventRecord(ev[l * C + c], sl);
    }
  }
  for (int l = 1; l < Le; ++l) { hipEventRecord(ev[Le * C + l], m->lay_s[l]); hipStreamWaitEvent(s0, ev[Le * C + l], 0); }
}

void encoder_forward(aocr_model* m, const Dims& d) {
  m->ctxa_fresh = false;                                             // new context
  hipStream_t s = m->s; const bool bf = m->bf16;
  const int B = d.B, T = d.T, He = m->He, Hd = m->Hd;
  const size_t slot = (size_t)B * He;
  int clG = 0, clRT = 0, clGroups = 0;
  const bool cluster = cluster_ok(m, B, T, clG, clRT, clGroups);
  const int pipeC = cluster ? layer_pipe_chunks(m, T, clG, clGroups) : 0;
  if (pipeC && layer_pipe_streams(m, (m->Le + 1) * pipeC + m->Le + 1)) {
    if (getenv("AOCR_TRACE")) fprintf(stderr, "[aocr] encoder forward: cluster kernels, layer wavefront of %d chunks\n", pipeC);
    encoder_forward_pipe(m, d, pipeC, clG, clRT, clGroups);
    if (m->context_b) copy2d_bf16(s, m->context, Hd, m->context_b, Hd, B * T, Hd);
    return;
  }
  for (int l = 0; l < m->Le; ++l) {
    ZeroList zl;
    for (int dir = 0; dir < 2; ++dir) {
      const LstmP& p = m->enc[dir][l];
      const float* xin = l == 0 ? m->X : m->ehs[dir][l - 1] + slot;        // Dropout(0) = identity (S6)
      const bf16_t* xinb = l == 0 ? m->Xb : (m->ehs_b[dir][l - 1] ? m->ehs_b[dir][l - 1] + slot : nullptr);
      prof_mark(m, AOCR_PROF_RNN_GEMM);
      if (l > 0 && m->drop_on) {                                           // LSTM.lua:68-69: x = Dropout(h of the layer below)
        if (cluster && xinb)                                               // the cluster kernels write h as bf16 only (the fp32 slots stay empty)
          dropout_apply_b(s, xinb, m->ehm[dir][l - 1], m->ehm_b[dir][l - 1], (int64_t)T * B * He, drop_site(m, (dir ? 48 : 32) + l + 1, 0));
        else
        dropout_apply(s, xin, m->ehm[dir][l - 1], m->ehm_b[dir][l - 1], (int64_t)T * B * He, drop_site(m, (dir ? 48 : 32) + l + 1, 0));
        xin = m->ehm[dir][l - 1]; xinb = m->ehm_b[dir][l - 1];
      }
      if (bf && xinb && p.swi.wb && p.in % 32 == 0)
        gemm_hh(s, xinb, p.in, p.swi.wb, p.in, m->ezx[dir][l], 4 * He, T * B, 4 * He, p.in, p.bi, p.bh, 0);
      else
        gemm(s, bf, xin, p.in, true, p.wi, p.in, true, m->ezx[dir][l], 4 * He, T * B, 4 * He, p.in, p.bi, p.bh, 0);
      // zero initial states of both directions (slot 0 / slot T+1): one launch for the layer instead of 6 memsets per direction
      zl.add(m->ehs[dir][l], slot * sizeof(float)); zl.add(m->ehs[dir][l] + (size_t)(T + 1) * slot, slot * sizeof(float));
      zl.add(m->ecs[dir][l], slot * sizeof(float)); zl.add(m->ecs[dir][l] + (size_t)(T + 1) * slot, slot * sizeof(float));
      if (m->ehs_b[dir][l]) {
        zl.add(m->ehs_b[dir][l], slot * sizeof(bf16_t)); zl.add(m->ehs_b[dir][l] + (size_t)(T + 1) * slot, slot * sizeof(bf16_t));
      }
    }
    prof_mark(m, AOCR_PROF_OTHER);
    zero_many(s, zl);
    prof_mark(m, AOCR_PROF_ENC_SEQ);
    const bool top = l == m->Le - 1;
    if (getenv("AOCR_TRACE")) fprintf(stderr, "[aocr] encoder layer %d forward: %s kernels\n", l, cluster ? "cluster" : seq_kernels_ok(m, B) ? "whole-sequence" : "per-step");
    if (cluster) {                                        // groups of He/64 CUs, recurrent weights resident in registers
      EncClFwdArgs a; a.B = B; a.T = T; a.He = He; a.Hd = Hd; a.groups = clGroups; a.epoch = next_epoch(m); a.xbuf = m->cl_xbuf; a.err = m->cl_err; a.xtab = m->cl_xtab;
      for (int dir = 0; dir < 2; ++dir) {
        EncSeqDir& e = a.d[dir];
        e.w = m->enc[dir][l].swh.wb; e.zx = m->ezx[dir][l]; e.hs = m->ehs[dir][l]; e.cs = m->ecs[dir][l]; e.hsb = m->ehs_b[dir][l];
        e.gates = m->egates[dir][l]; e.ctx = top ? m->context + dir * He : nullptr; e.reverse = dir;
      }
      enc_cluster_forward(s, a, clG, clRT, comm_reserved_cus(m));
      continue;
    }
    if (seq_kernels_ok(m, B)) {                         // whole-sequence kernel: one launch for all T steps of both directions
      EncSeqFwdArgs a; a.B = B; a.T = T; a.He = He; a.Hd = Hd;
      for (int dir = 0; dir < 2; ++dir) {
        EncSeqDir& e = a.d[dir];
        e.w = m->enc[dir][l].swh.wb; e.zx = m->ezx[dir][l]; e.hs = m->ehs[dir][l]; e.cs = m->ecs[dir][l]; e.hsb = m->ehs_b[dir][l];
        e.gates = m->egates[dir][l]; e.ctx = top ? m->context + dir * He : nullptr; e.reverse = dir;
      }
      enc_seq_forward(s, a);
      continue;
    }
    for (int i = 0; i < T; ++i) {
      LoadK la[2]; LoadKh2 lah[2]; EpGatesFwd ee[2]; const ShW* w0[2]; const ShW* w1[2] = {nullptr, nullptr};
      for (int dir = 0; dir < 2; ++dir) {
        const LstmP& p = m->enc[dir][l];
        const int t = dir == 0 ? i : T - 1 - i;
        const int prev = dir == 0 ? t : t + 2;
        float* hs = m->ehs[dir][l]; float* cs = m->ecs[dir][l];
        la[dir] = make_loadk(hs + prev * slot, He, B, He);
        lah[dir] = make_loadkh(m->ehs_b[dir][l] ? m->ehs_b[dir][l] + prev * slot : nullptr, He, B, He);
        w0[dir] = &p.swh;
        EpGatesFwd& e = ee[dir];
        e.zx = m->ezx[dir][l] + (size_t)t * B * 4 * He; e.ldzx = 4 * He; e.b1 = nullptr; e.b2 = nullptr;
        e.c_prev = cs + prev * slot; e.ldcp = He;
        e.c_out = cs + (size_t)(t + 1) * slot; e.ldc = He; e.h_out = hs + (size_t)(t + 1) * slot; e.ldh = He;
        e.h_out2 = top ? m->context + (size_t)t * Hd + dir * He : nullptr; e.ldh2 = (int64_t)T * Hd;   // model.lua:303,315
        e.gates = m->egates[dir][l] + (size_t)t * B * 4 * He; e.ldg = 4 * He; e.M = B; e.H = He;
        if (m->ehs_b[dir][l]) { e.hb = m->ehs_b[dir][l] + (size_t)(t + 1) * slot; e.ldhb = He; }
      }
      run_gates_fwd(m, 2, la, w0, w1, ee, B, He, lah);
    }
  }
  // bf16 mode: the attention kernels of the decoder (2 x L launches per step) read the context from a bf16 shadow
  prof_mark(m, AOCR_PROF_OTHER);
  if (m->context_b) copy2d_bf16(s, m->context, Hd, m->context_b, Hd, B * T, Hd);
}

// BPTT through both encoder directions, model.lua:662-690.  On entry dc_st[0] / dh_rec[0] of the decoder hold
// d c1(0) / d h1(0) (quirk S5: d h1(0) is passed on even when h1(0) was zeroed in the forward pass).
// the layer wavefront on the way back (see encoder_forward_pipe): the top layer on the main stream, layer l on lay_s[Le-1-l]; layer l
// starts chunk c when layer l+1's d x of that chunk (GEMM over the chunk's rows) is complete.  The weight gradients (and layer 0's d x)
// need the whole sequence: behind the layer's last chunk, on its stream.
static void encoder_backward_pipe(aocr_model* m, const Dims& d, int C, int clG, int clRT, int clGroups) {
  hipStream_t s0 = m->s;
  const int B = d.B, T = d.T, He = m->He, Hd = m->Hd, Le = m->Le;
  const size_t slot = (size_t)B * He;
  for (int dir = 0; dir < 2; ++dir)
    for (int l = 0; l < Le; ++l) {
      if (l == Le - 1) copy2d(s0, m->dc_st[0] + dir * He, Hd, m->edc[dir][l], He, B, He);        // model.lua:666,680
      else hipMemsetAsync(m->edc[dir][l], 0, slot * sizeof(float), s0);
    }
  reserve_epochs(m, (unsigned)(C * Le));
  hipEvent_t* ev = m->lay_ev.data();
  hipEventRecord(ev[Le * C], s0);
  for (int l = 1; l < Le; ++l) hipStreamWaitEvent(m->lay_s[l], ev[Le * C], 0);
  auto stream_of = [&](int l) { return l == Le - 1 ? s0 : m->lay_s[Le - 1 - l]; };
  const int Tc = (T + C - 1) / C;
  for (int c = 0; c < C; ++c) {
    const int it0 = c * Tc, it1 = std::min(T, it0 + Tc);
    if (it0 >= it1) break;
    for (int l = Le - 1; l >= 0; --l) {
      hipStream_t sl = stream_of(l);
      const bool top = l == Le - 1;
      if (!top) hipStreamWaitEvent(sl, ev[(l + 1) * C + c], 0);
      EncClBwdArgs a; a.B = B; a.T = T; a.He = He; a.groups = clGroups; a.epoch = next_epoch(m); a.pbuf = m->cl_pbuf; a.err = m->cl_err; a.xtab = m->cl_xtab;
      a.it0 = it0; a.it1 = it1; a.gslot = l * 2 * clGroups;
      for (int dir = 0; dir < 2; ++dir) {
        EncSeqBwdDir& e = a.d[dir];
        e.wt = m->enc[dir][l].swh.wtb;
        if (top) { e.dh1 = m->dctx + dir * He; e.dh1_row = (int64_t)T * Hd; e.dh1_t = Hd; }      // model.lua:670,684
        else { e.dh1 = m->edxl[dir]; e.dh1_row = He; e.dh1_t = (int64_t)slot; }
        e.dh2 = top ? m->dh_rec[0] + dir * He : nullptr; e.dh2_row = Hd;                            // model.lua:667,681
        e.dc = m->edc[dir][l]; e.gates = m->egates[dir][l]; e.cs = m->ecs[dir][l];
        e.dz = m->edz[dir][l]; e.dzb = m->edz_b[dir][l]; e.forward_dir = dir == 0;
        e.dbi = m->enc[dir][l].dbi; e.dbh = m->enc[dir][l].dbh;
      }
      enc_cluster_backward(sl, a, clG, clRT, comm_reserved_cus(m), Le);
      if (l > 0) {
        for (int dir = 0; dir < 2; ++dir) {                // d x of this chunk's steps: t = T-1-it (the fw direction's BPTT) / it
          const LstmP& p = m->enc[dir][l];
          const int t_lo = dir == 0 ? T - it1 : it0, n = it1 - it0;
          float* dxo = m->edxl[dir] + (size_t)t_lo * slot;
          gemm_hh(sl, m->edz_b[dir][l] + (size_t)t_lo * B * 4 * He, 4 * He, p.swi.wtb, 4 * He, dxo, p.in, n * B, p.in, 4 * He, nullptr, nullptr, 0);
          if (m->drop_on) dropout_apply(sl, dxo, dxo, nullptr, (int64_t)n * slot, drop_site(m, (dir ? 48 : 32) + l + 1, (long long)t_lo * (long long)slot));     // Dropout backward
        }
        hipEventRecord(ev[l * C + c], sl);
      }
    }
  }
  for (int l = Le - 1; l >= 0; --l) {
    hipStream_t sl = stream_of(l);
    WGradProblem wg[4]; int nwg = 0;
    for (int dir = 0; dir < 2; ++dir) {
      const LstmP& p = m->enc[dir][l];
      const float* xin = l == 0 ? m->X : m->ehs[dir][l - 1] + slot;
      const float* hprev = m->ehs[dir][l] + (dir == 0 ? 0 : 2 * slot);
      const bf16_t* dzb = m->edz_b[dir][l];
      const bf16_t* xinb = l == 0 ? m->Xb : m->ehs_b[dir][l - 1] + slot;
      const bf16_t* hprevb = m->ehs_b[dir][l] + (dir == 0 ? 0 : 2 * slot);
      if (l > 0 && m->drop_on) { xin = m->ehm[dir][l - 1]; xinb = m->ehm_b[dir][l - 1]; }          // the layer saw the masked input
      wg[nwg++] = WGradProblem{m->edz[dir][l], 4 * He, xin, p.in, p.dwi, p.in, 4 * He, p.in, T * B, dzb, xinb};
      wg[nwg++] = WGradProblem{m->edz[dir][l], 4 * He, hprev, He, p.dwh, He, 4 * He, He, T * B, dzb, hprevb};
      if (l == 0) gemm_hh(sl, dzb, 4 * He, p.swi.wtb, 4 * He, m->dX, p.in, T * B, p.in, 4 * He, nullptr, nullptr, dir == 1 ? EP_ACCUM : 0);   // model.lua:675 copy, :689 add
    }
    grouped_wgrad(sl, true, wg, nwg);
  }
  for (int l = 1; l < Le; ++l) { hipEventRecord(ev[Le * C + l], m->lay_s[l]); hipStreamWaitEvent(s0, ev[Le * C + l], 0); }
}

static void encoder_backward(aocr_model* m, const Dims& d) {
  hipStream_t s = m->s; const bool bf = m->bf16;
  const int B = d.B, T = d.T, He = m->He, Hd = m->Hd;
  const size_t slot = (size_t)B * He;
  {
    int clG = 0, clRT = 0, clGroups = 0;
    bool cluster = cluster_ok(m, B, T, clG, clRT, clGroups);
    for (int l = 0; l < m->Le && cluster; ++l) cluster = m->edz_b[0][l] && m->enc[0][l].swh.wtb && m->enc[0][l].swi.wtb;
    const int pipeC = cluster ? layer_pipe_chunks(m, T, clG, clGroups) : 0;
    if (pipeC && layer_pipe_streams(m, (m->Le + 1) * pipeC + m->Le + 1)) {
      if (getenv("AOCR_TRACE")) fprintf(stderr, "[aocr] encoder backward: cluster kernels, layer wavefront of %d chunks\n", pipeC);
      encoder_backward_pipe(m, d, pipeC, clG, clRT, clGroups);
      return;
    }
  }
  for (int l = m->Le - 1; l >= 0; --l) {
    const bool top = l == m->Le - 1;
    prof_mark(m, AOCR_PROF_ENC_SEQ);
    int clG = 0, clRT = 0, clGroups = 0;
    const bool cluster = cluster_ok(m, B, T, clG, clRT, clGroups) && m->edz_b[0][l] && m->enc[0][l].swh.wtb;
    const bool dc_direct = top && cluster && !env_on("AOCR_ENC_DC_COPY");      // the cluster kernel reads the halves of the decoder's initial-state gradient in place (one 9 us launch off the main stream's critical chain)
    if (top && !dc_direct) copy2d_pair(s, m->dc_st[0], m->dc_st[0] + He, Hd, m->edc[0][l], m->edc[1][l], He, B, He);        // model.lua:666,680 (both directions in one launch)
    else if (!top) for (int dir = 0; dir < 2; ++dir) hipMemsetAsync(m->edc[dir][l], 0, slot * sizeof(float), s);
    const bool seq = cluster || (seq_kernels_ok(m, B) && m->edz_b[0][l] && m->enc[0][l].swh.wtb);
    if (getenv("AOCR_TRACE")) fprintf(stderr, "[aocr] encoder layer %d backward: %s kernels\n", l, cluster ? "cluster" : seq ? "whole-sequence" : "per-step");
    if (cluster) {
      EncClBwdArgs a; a.B = B; a.T = T; a.He = He; a.groups = clGroups; a.epoch = next_epoch(m); a.pbuf = m->cl_pbuf; a.err = m->cl_err; a.xtab = m->cl_xtab;
      for (int dir = 0; dir < 2; ++dir) {
        EncSeqBwdDir& e = a.d[dir];
        e.wt = m->enc[dir][l].swh.wtb;
        if (top) { e.dh1 = m->dctx + dir * He; e.dh1_row = (int64_t)T * Hd; e.dh1_t = Hd; }      // model.lua:670,684
        else { e.dh1 = m->edxl[dir]; e.dh1_row = He; e.dh1_t = (int64_t)slot; }
        e.dh2 = top ? m->dh_rec[0] + dir * He : nullptr; e.dh2_row = Hd;                            // model.lua:667,681
        e.dc = m->edc[dir][l]; e.gates = m->egates[dir][l]; e.cs = m->ecs[dir][l];
        if (dc_direct) { e.dc_in = m->dc_st[0] + dir * He; e.dc_in_row = Hd; }
        e.dz = m->edz[dir][l]; e.dzb = m->edz_b[dir][l]; e.forward_dir = dir == 0;
        e.dbi = m->enc[dir][l].dbi; e.dbh = m->enc[dir][l].dbh;
      }
      enc_cluster_backward(s, a, clG, clRT, comm_reserved_cus(m));
    } else if (seq) {
      EncSeqBwdArgs a; a.B = B; a.T = T; a.He = He;
      for (int dir = 0; dir < 2; ++dir) {
        EncSeqBwdDir& e = a.d[dir];
        e.wt = m->enc[dir][l].swh.wtb;
        if (top) { e.dh1 = m->dctx + dir * He; e.dh1_row = (int64_t)T * Hd; e.dh1_t = Hd; }      // model.lua:670,684
        else { e.dh1 = m->edxl[dir]; e.dh1_row = He; e.dh1_t = (int64_t)slot; }
        e.dh2 = top ? m->dh_rec[0] + dir * He : nullptr; e.dh2_row = Hd;                            // model.lua:667,681
        e.dc = m->edc[dir][l]; e.gates = m->egates[dir][l]; e.cs = m->ecs[dir][l];
        e.dz = m->edz[dir][l]; e.dzb = m->edz_b[dir][l]; e.forward_dir = dir == 0;
      }
      enc_seq_backward(s, a);
    }
    for (int i = 0; i < T && !seq; ++i) {
      LoadK la[2]; LoadKh2 lah[2]; EpGatesBwd ee[2]; const ShW* ww[2];
      for (int dir = 0; dir < 2; ++dir) {
        const LstmP& p = m->enc[dir][l];
        const int t = dir == 0 ? T - 1 - i : i;
        const int tn = dir == 0 ? t + 1 : t - 1;                                     // step processed just before
        const int prev = dir == 0 ? t : t + 2;
        float* dz = m->edz[dir][l];
        la[dir] = make_loadk(i == 0 ? dz : dz + (size_t)tn * B * 4 * He, 4 * He, B, i == 0 ? 0 : 4 * He);
        lah[dir] = make_loadkh((i == 0 || !m->edz_b[dir][l]) ? nullptr : m->edz_b[dir][l] + (size_t)tn * B * 4 * He, 4 * He, B, i == 0 ? 0 : 4 * He);
        ww[dir] = &p.swh;
        EpGatesBwd& e = ee[dir];
        if (top) { e.dh1 = m->dctx + (size_t)t * Hd + dir * He; e.ld1 = (int64_t)T * Hd; }    // model.lua:670,684
        else { e.dh1 = m->edxl[dir] + (size_t)t * slot; e.ld1 = He; }
        e.dh2 = (top && i == 0) ? m->dh_rec[0] + dir * He : nullptr; e.ld2 = Hd;               // model.lua:667,681
        e.dc_in = m->edc[dir][l]; e.lddc = He;
        e.gates = m->egates[dir][l] + (size_t)t * B * 4 * He; e.ldg = 4 * He;
        e.c_prev = m->ecs[dir][l] + prev * slot; e.ldcp = He; e.c = m->ecs[dir][l] + (size_t)(t + 1) * slot; e.ldcc = He;
        e.dz = dz + (size_t)t * B * 4 * He; e.lddz = 4 * He; e.dc_out = m->edc[dir][l]; e.lddco = He; e.M = B; e.H = He;
        if (m->edz_b[dir][l]) { e.dzb = m->edz_b[dir][l] + (size_t)t * B * 4 * He; e.lddzb = 4 * He; }
      }
      run_gates_bwd(m, 2, la, ww, ee, B, He, lah);
    }
    prof_mark(m, AOCR_PROF_RNN_GEMM);
    // round 4: the layer's weight gradients join the decoder's on the side stream (nothing reads them before the optimizer), behind an event that
    // marks the end of this layer's BPTT: 0.11 ms off the main stream at C3, where they ran between d X and the CNN backward pass
    bool wg_side = false;
    if (m->side_busy && m->side && !env_on("AOCR_NO_ENC_WGRAD_SIDE")) {
      if (!m->enc_ev && hipEventCreateWithFlags(&m->enc_ev, hipEventDisableTiming) != hipSuccess) m->enc_ev = nullptr;
      if (m->enc_ev) { hipEventRecord(m->enc_ev, s); wg_side = true; }
    }
    WGradProblem wg[4]; int nwg = 0;
    // layer 0: d X = d z_fw W_i2h_fw + d z_bw W_i2h_bw (model.lua:675 copy, :689 add) as ONE product over K = 2 x 4He (round 4: one launch instead of two,
    // the second of which re-read the first's 33 MB output to add to it)
    bool dx_cat = false;
    if (l == 0 && bf && !m->drop_on && m->edz_b[0][0] && m->edz_b[1][0] && m->enc[0][0].swi.wtb && m->enc[1][0].swi.wtb)
      dx_cat = gemm_hh_cat(s, m->edz_b[0][0], m->edz_b[1][0], 4 * He, m->enc[0][0].swi.wtb, m->enc[1][0].swi.wtb, 4 * He, m->dX, m->enc[0][0].in, T * B, m->enc[0][0].in, 4 * He, 4 * He);
    for (int dir = 0; dir < 2; ++dir) {
      const LstmP& p = m->enc[dir][l];
      const float* xin = l == 0 ? m->X : m->ehs[dir][l - 1] + slot;
      const float* hprev = m->ehs[dir][l] + (dir == 0 ? 0 : 2 * slot);
      const float* dz = m->edz[dir][l];
      const bf16_t* dzb = m->edz_b[dir][l];
      const bf16_t* xinb = l == 0 ? m->Xb : (m->ehs_b[dir][l - 1] ? m->ehs_b[dir][l - 1] + slot : nullptr);
      const bf16_t* hprevb = m->ehs_b[dir][l] ? m->ehs_b[dir][l] + (dir == 0 ? 0 : 2 * slot) : nullptr;
      if (l > 0 && m->drop_on) { xin = m->ehm[dir][l - 1]; xinb = m->ehm_b[dir][l - 1]; }          // the layer saw the masked input
      wg[nwg++] = WGradProblem{dz, 4 * He, xin, p.in, p.dwi, p.in, 4 * He, p.in, T * B, dzb, xinb};
      wg[nwg++] = WGradProblem{dz, 4 * He, hprev, He, p.dwh, He, 4 * He, He, T * B, dzb, hprevb};
      if (!cluster) colsum_accum(s, dz, 4 * He, (int64_t)T * B, 4 * He, p.dbi, p.dbh);        // both biases see the same d z (LSTM.lua:79-88); the cluster kernel sums them itself
      float* dxo = l == 0 ? m->dX : m->edxl[dir];
      const int dxf = (l == 0 && dir == 1) ? EP_ACCUM : 0;                          // model.lua:675 copy, :689 add
      if (dx_cat) { /* done above */ }
      else if (bf && dzb && p.swi.wtb) gemm_hh(s, dzb, 4 * He, p.swi.wtb, 4 * He, dxo, p.in, T * B, p.in, 4 * He, nullptr, nullptr, dxf);
      else if (p.swi.wtf) gemm(s, bf, dz, 4 * He, true, p.swi.wtf, 4 * He, true, dxo, p.in, T * B, p.in, 4 * He, nullptr, nullptr, dxf);
      else gemm(s, bf, dz, 4 * He, true, p.wi, p.in, false, dxo, p.in, T * B, p.in, 4 * He, nullptr, nullptr, dxf);
      if (l > 0 && m->drop_on) dropout_apply(s, dxo, dxo, nullptr, (int64_t)T * B * He, drop_site(m, (dir ? 48 : 32) + l + 1, 0));     // Dropout backward
    }
    if (wg_side) { hipStreamWaitEvent(m->side, m->enc_ev, 0); grouped_wgrad(m->side, bf, wg, nwg, m->wg_part, m->wg_part_floats); }
    else grouped_wgrad(s, bf, wg, nwg, m->side_busy ? nullptr : m->wg_part, m->wg_part_floats);      // (the side stream may be using the slab scratch: the decoder's weight gradients)
  }
}

// ------------------------------------------------------------------------------------------------
// decoder
// ------------------------------------------------------------------------------------------------
struct DecStepIO {
  int R, ctx_div;
  const float* zx1; const float* feed;
  const int32_t* zx_tok = nullptr; int64_t zx_tok_stride = 1;   // zx1 is a per-token table (decode): row r uses table row zx_tok[r*stride]-1
  const float* c_prev[MAXL]; const float* h_prev[MAXL];
  float* c_new[MAXL]; float* h_new[MAXL]; float* gates[MAXL];
  float *q, *a, *cat, *out;
  const bf16_t* ctxa = nullptr;                                 // bf16(ctx W_a): scores against it, no q = W_a h launch (training chain at Hd = 1024, T <= 64)
  // bf16 shadows (teacher-forced path in bf16 mode; nullptr otherwise)
  const bf16_t* feed_b = nullptr; const bf16_t* hb_prev[MAXL] = {nullptr, nullptr, nullptr, nullptr};
  bf16_t* hb_new[MAXL] = {nullptr, nullptr, nullptr, nullptr}; bf16_t* cat_b = nullptr; bf16_t* out_b = nullptr;
  // dropout (training): masked copies of h_new[l] for the layer above, and the mask of the attention output
  float* hm_new[MAXL] = {nullptr, nullptr, nullptr, nullptr}; bf16_t* hmb_new[MAXL] = {nullptr, nullptr, nullptr, nullptr};
  DropSpec drop_h[MAXL]; DropSpec drop_out;
};

// one decoder clone forward, LSTM.lua:18-122: LSTM layers, then attention (LSTM.lua:124-162).
static void dec_step_forward(aocr_model* m, const DecStepIO& io, int T) {
  hipStream_t s = m->s; const bool bf = m->bf16;
  const int R = io.R, Hd = m->Hd, E = m->E;
  for (int l = 0; l < m->Ld; ++l) {
    const LstmP& p = m->dec[l];
    LoadK la; LoadKh2 lah; EpGatesFwd e; const ShW* w0; const ShW* w1 = nullptr;
    const bool sh = io.hb_new[0] != nullptr;
    if (l == 0) {
      if (m->cfg.input_feed) {
        la = make_loadk2(io.feed, Hd, Hd, io.h_prev[0], Hd, Hd, R); w0 = &p.swi; w1 = &p.swh;
        lah = make_loadkh2(sh ? io.feed_b : nullptr, Hd, Hd, io.hb_prev[0], Hd, Hd, R);
      } else { la = make_loadk(io.h_prev[0], Hd, R, Hd); w0 = &p.swh; lah = make_loadkh(sh ? io.hb_prev[0] : nullptr, Hd, R, Hd); }
      e.zx = io.zx1; e.ldzx = 4 * Hd; e.b1 = nullptr; e.b2 = nullptr; e.zx_tok = io.zx_tok; e.zx_tok_stride = io.zx_tok_stride;
    } else {
      const float* xl = io.hm_new[l - 1] ? io.hm_new[l - 1] : io.h_new[l - 1];        // LSTM.lua:68-69: Dropout on the input of the layers above the first
      const bf16_t* xlb = io.hm_new[l - 1] ? io.hmb_new[l - 1] : io.hb_new[l - 1];
      la = make_loadk2(xl, Hd, Hd, io.h_prev[l], Hd, Hd, R); w0 = &p.swi; w1 = &p.swh;
      lah = make_loadkh2(sh ? xlb : nullptr, Hd, Hd, io.hb_prev[l], Hd, Hd, R);
      e.zx = nullptr; e.ldzx = 0; e.b1 = p.bi; e.b2 = p.bh;
    }
    e.c_prev = io.c_prev[l]; e.ldcp = Hd; e.c_out = io.c_new[l]; e.ldc = Hd; e.h_out = io.h_new[l]; e.ldh = Hd;
    const bool top = l == m->Ld - 1;
    e.h_out2 = top ? io.cat + Hd : nullptr; e.ldh2 = 2 * Hd;                         // JoinTable [c ; h_top], LSTM.lua:153
    e.gates = io.gates[l]; e.ldg = 4 * Hd; e.M = R; e.H = Hd;
    if (sh) { e.hb = io.hb_new[l]; e.ldhb = Hd; if (top) { e.hb2 = io.cat_b + Hd; e.ldhb2 = 2 * Hd; } }
    if (!top && io.hm_new[l]) { e.h_out2 = io.hm_new[l]; e.ldh2 = Hd; e.hb2 = sh ? io.hmb_new[l] : nullptr; e.ldhb2 = Hd; e.drop = io.drop_h[l]; }
    run_gates_fwd(m, 1, &la, &w0, &w1, &e, R, Hd, &lah);
  }
  {
    const bool sh = io.hb_new[0] != nullptr;
    if (io.ctxa && attention_dual_ok(T, Hd, m->context_b, io.ctxa))
      attention_forward_dual(s, io.h_new[m->Ld - 1], Hd, io.a, io.cat, 2 * Hd, R, T, io.ctx_div, io.cat_b, 2 * Hd, m->context_b, io.ctxa);      // ctx[t] . (W_a h) = (ctx W_a)[t] . h
    else {
      LoadKh2 qa = make_loadkh(sh ? io.hb_new[m->Ld - 1] : nullptr, Hd, R, Hd);
      run_store_nt(m, make_loadk(io.h_new[m->Ld - 1], Hd, R, Hd), m->swa, make_store(io.q, Hd, R, Hd), R, &qa);      // q = W_a h_top, LSTM.lua:131
      attention_forward(s, m->context, io.q, io.a, io.cat, 2 * Hd, R, T, Hd, io.ctx_div, io.cat_b, 2 * Hd, m->context_b);
    }
    EpStore eo = make_store(io.out, Hd, R, Hd, nullptr, nullptr, EP_TANH);
    eo.Cb = io.out_b; eo.ldcb = Hd;
    LoadKh2 ca = make_loadkh(sh ? io.cat_b : nullptr, 2 * Hd, R, 2 * Hd);
    run_store_nt(m, make_loadk(io.cat, 2 * Hd, R, 2 * Hd), m->swc, eo, R, &ca);                                       // LSTM.lua:155
    if (io.drop_out.thr != 0) dropout_apply(s, io.out, io.out, io.out_b, (int64_t)R * Hd, io.drop_out);                // LSTM.lua:116-118 (training with p > 0 only)
  }
  (void)E; (void)bf;
}

// initial decoder state from the encoder's final states, model.lua:539-552 (+ quirk S5)
static void dec_init_state(aocr_model* m, const Dims& d, float* const* c0, float* const* h0, float* feed0, int R, bool shadows = false,
                           bf16_t* const* hb0 = nullptr, bf16_t* feedb0 = nullptr) {      // hb0 / feedb0: bf16 copies of the initial state somewhere else than the training buffers (decode chain)
  const int B = d.B, T = d.T, He = m->He, Hd = m->Hd; const size_t slot = (size_t)B * He;
  (void)R;
  const int lt = m->Le - 1;
  DecInitArgs a{};
  for (int l = 0; l < m->Ld && l < 4; ++l) { a.c0[l] = c0[l]; a.h0[l] = h0[l]; a.hb[l] = hb0 ? hb0[l] : (shadows ? m->dhs_b[l] : nullptr); }
  a.feed0 = feed0; a.outb = feedb0 ? feedb0 : (shadows ? m->out_b : nullptr);
  // c1(0) = [c_fw(T) ; c_bw(1)]; model.lua:549-552 (quirk S5) zeroes h1(0) instead of h2(0) with input feed and two or more layers
  a.cfw = m->ecs[0][lt] + (size_t)T * slot; a.cbw = m->ecs[1][lt] + (size_t)1 * slot;
  a.hfw = m->ehs[0][lt] + (size_t)T * slot; a.hbw = m->ehs[1][lt] + (size_t)1 * slot;
  a.B = B; a.He = He; a.Hd = Hd; a.Ld = m->Ld; a.copy_h = (m->cfg.input_feed && m->Ld >= 2) ? 0 : 1;
  dec_init(m->s, a);
}

// The decoder cluster kernel (dec_cluster.hip): bf16 mode, Hd = 512, two layers, input feed (the reference's defaults at
// He = 256); AOCR_NO_DEC_CLUSTER=1 keeps the per-step launch chain (parity tests compare the two).
static bool dec_cluster_ok(const aocr_model* m, int T, int L) {
  if (!m->bf16 || !m->dc_xbuf || !m->ctxa_b || !m->dec[0].swi.wb || !m->swa.wtb || !m->swc.wb) return false;
  const char* e = getenv("AOCR_NO_DEC_CLUSTER");
  if (e && e[0] == '1') return false;
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return dec_cluster_supported(m->Hd, m->Ld, m->cfg.input_feed, T, L, cus);
}

static bool dec_cluster_bwd_ok(const aocr_model* m, int T, int L) {
  if (!m->dc_bxbuf || !m->dpre_b || !m->dq_b || !m->ddz_b[0] || !m->dec[0].swi.wtb || !m->swc.wtb) return false;
  const char* e = getenv("AOCR_NO_DEC_CLUSTER_BWD");
  if (e && e[0] == '1') return false;
  static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
  return dec_cluster_bwd_supported(m->Hd, m->Ld, m->cfg.input_feed, T, L, cus);
}

static bool side_create(aocr_model* m);
// teacher-forced decoder loop, model.lua:553-568 (train) / :604-627 (gold pass); the projector (model.lua:560)
// is hoisted out of the loop: logits for all L steps in one contraction.
void decoder_tf_forward(aocr_model* m, const Dims& d, const int32_t* tgt, int64_t st, int64_t sb, bool keep_gates) {
  hipStream_t s = m->s; const bool bf = m->bf16;
  const int B = d.B, T = d.T, L = d.L, Hd = m->Hd, E = m->E;
  const size_t slot = (size_t)B * Hd;
  prof_mark(m, AOCR_PROF_RNN_GEMM);
  const LstmP& p1 = m->dec[0];
  const bool sh = m->bf16 && m->out_b != nullptr;
  const bool drop = keep_gates && m->drop_on;                       // training only
  const bool drop_cl = drop && m->dhm_b[0] && !getenv("AOCR_NO_DEC_CLUSTER_DROP");     // round 3: the cluster kernels evaluate the masks themselves
  const bool use_cl = sh && (!drop || drop_cl) && dec_cluster_ok(m, T, L);
  // Round 4: the embedding part of the first layer's gate input depends on the token only (nn.LookupTable -> W_i2h[:, :E], LSTM.lua:55-56,79-80).
  // The whole-sequence kernel reads it from the per-token table [V][4 Hd] (what the decode path already does) instead of a (L B, 4 Hd) tensor
  // produced by a gather + a K = E product: the same dot products (same kernel, same k order: bit-identical rows), 39 rows instead of 6144 at C3,
  // and 50 MB less written and read per step.  The backward pass follows (decoder_backward: sums of d z by token).  AOCR_NO_EMB_TABLE=1: the tensor.
  // Round 6: the launch chain of bf16 mode takes the table too (its gate epilogue gathers the token's row, as the decode path always did): at the reference-default
  // shape the (L B, 4 Hd) tensor was a 157 MB product in front of the loop and a K = 4 Hd -> E product + scatter behind it (87 + 99 us of a 8.3 ms step).
  m->emb_table = (use_cl || (sh && !env_on("AOCR_NO_EMB_TABLE_CHAIN"))) && segsum_supported(4 * Hd, m->V, E) && !env_on("AOCR_NO_EMB_TABLE");
  if (m->tab_ready) { hipStreamWaitEvent(s, m->tab_done, 0); m->tab_ready = false; m->tab_valid = true; }       // step_prologue of this call (joined even when unused: the decode path writes the same buffer)
  if (m->emb_table) { if (!m->tab_valid) { gemm(s, bf, m->lookup, E, true, p1.wi, p1.in, true, m->bzx_tab, 4 * Hd, m->V, 4 * Hd, E, p1.bi, p1.bh, 0); m->tab_valid = true; } }      // (tab_valid: this API call already has the table -- the beam pass in front of a gold pass)
  else {
    embedding_gather(s, m->lookup, tgt, st, sb, m->emb_all, L, B, E);
    gemm(s, bf, m->emb_all, E, true, p1.wi, p1.in, true, m->zx1_all, 4 * Hd, L * B, 4 * Hd, E, p1.bi, p1.bh, 0);
  }
  float* c0[MAXL]; float* h0[MAXL];
  for (int l = 0; l < m->Ld; ++l) { c0[l] = m->dcs[l]; h0[l] = m->dhs[l]; }
  prof_mark(m, AOCR_PROF_DEC_FWD);
  dec_init_state(m, d, c0, h0, m->out_all, B, sh);
  m->dgates_il = false;
  if (use_cl) {
    // scores against the pre-multiplied context: ctx[t] . (W_a h) = (ctx W_a)[t] . h, LSTM.lua:131-137
    if (!m->ctxa_fresh) gemm_hh_shadow(s, m->context_b, Hd, m->swa.wtb, Hd, m->dctx, Hd, m->ctxa_b, Hd, B * T, Hd, Hd);
    m->ctxa_fresh = false;
    DecClFwdArgs a; a.B = B; a.T = T; a.L = L; a.epoch = next_epoch(m);
    a.w1i = m->dec[0].swi.wb; a.w1h = m->dec[0].swh.wb; a.w2i = m->dec[1].swi.wb; a.w2h = m->dec[1].swh.wb; a.wc = m->swc.wb;
    a.b2i = m->dec[1].bi; a.b2h = m->dec[1].bh; a.zx1 = m->zx1_all; a.ctxb = m->context_b; a.ctxa = m->ctxa_b;
    if (m->emb_table) { a.zx1 = m->bzx_tab; a.zx_tok = tgt; a.zx_st = st; a.zx_sb = sb; a.V = m->V; }
    for (int l = 0; l < 2; ++l) { a.cs[l] = m->dcs[l]; a.hsb[l] = m->dhs_b[l]; a.gates[l] = keep_gates ? m->dgates[l] : nullptr; }
    a.a_all = m->a_all; a.out = m->out_all; a.cat_b = m->cat_b; a.out_b = m->out_b;
    a.xbuf = m->dc_xbuf; a.xtab = m->dc_xtab; a.err = m->cl_err;
    if (drop) { a.drop_h = drop_site(m, 2, 0); a.drop_out = drop_site(m, 16, 0); a.hm_b = m->dhm_b[0]; }     // (.off = the step's offset, added in the kernel)
    dec_cluster_forward(s, a);
    m->dgates_il = true;
    if (keep_gates) {                                              // q = W_a h_top for all L steps: only the backward pass reads it (attention_dctx, behind the decoder BPTT)
      // round 5: on the side stream, beside the projector / loss / d logits launches that follow (small grids: 85 us of mostly idle chip between the two
      // whole-sequence kernels at C3), joined in front of attention_dctx -- 21 us off the main stream
      auto ev = [](hipEvent_t& e) { return e || hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
      if (!m->prof_on && !getenv("AOCR_NO_SIDE_WGRAD") && !env_on("AOCR_NO_Q_SIDE") && side_create(m) && ev(m->q_go) && ev(m->q_done)) {
        hipEventRecord(m->q_go, s); hipStreamWaitEvent(m->side, m->q_go, 0);
        gemm_hh(m->side, m->dhs_b[1] + slot, Hd, m->swa.wb, Hd, m->q_all, Hd, L * B, Hd, Hd, nullptr, nullptr, 0);
        hipEventRecord(m->q_done, m->side); m->q_pending = true;
      } else gemm_hh(s, m->dhs_b[1] + slot, Hd, m->swa.wb, Hd, m->q_all, Hd, L * B, Hd, Hd, nullptr, nullptr, 0);
    }
  } else {
  // the chain at Hd = 1024 scores against the pre-multiplied context as well (bf16 shadows, T <= 64): one hoisted (B T, Hd) x (Hd, Hd) product instead of L launches of q = W_a h
  const bool chain_ctxa = sh && m->ctxa_b && m->swa.wtb && attention_dual_ok(T, Hd, m->context_b, m->ctxa_b);
  if (chain_ctxa && !m->ctxa_fresh) gemm_hh_shadow(s, m->context_b, Hd, m->swa.wtb, Hd, m->dctx, Hd, m->ctxa_b, Hd, B * T, Hd, Hd);
  m->ctxa_fresh = false;
  for (int t = 0; t < L; ++t) {
    DecStepIO io; io.R = B; io.ctx_div = 1;
    if (chain_ctxa) io.ctxa = m->ctxa_b;
    io.zx1 = m->zx1_all + (size_t)t * B * 4 * Hd; io.feed = m->out_all + (size_t)t * slot;
    if (m->emb_table) { io.zx1 = m->bzx_tab; io.zx_tok = tgt + (int64_t)t * st; io.zx_tok_stride = sb; }      // row b of step t reads the table row of its token
    for (int l = 0; l < m->Ld; ++l) {
      io.c_prev[l] = m->dcs[l] + (size_t)t * slot; io.h_prev[l] = m->dhs[l] + (size_t)t * slot;
      io.c_new[l] = m->dcs[l] + (size_t)(t + 1) * slot; io.h_new[l] = m->dhs[l] + (size_t)(t + 1) * slot;
      io.gates[l] = keep_gates ? m->dgates[l] + (size_t)t * B * 4 * Hd : nullptr;
    }
    io.q = m->q_all + (size_t)t * slot; io.a = m->a_all + (size_t)t * B * T; io.cat = m->cat_all + (size_t)t * B * 2 * Hd;
    io.out = m->out_all + (size_t)(t + 1) * slot;
    if (sh) {
      io.feed_b = m->out_b + (size_t)t * slot; io.out_b = m->out_b + (size_t)(t + 1) * slot; io.cat_b = m->cat_b + (size_t)t * B * 2 * Hd;
      for (int l = 0; l < m->Ld; ++l) { io.hb_prev[l] = m->dhs_b[l] + (size_t)t * slot; io.hb_new[l] = m->dhs_b[l] + (size_t)(t + 1) * slot; }
    }
    if (drop) {
      for (int l = 0; l + 1 < m->Ld; ++l) {
        io.hm_new[l] = m->dhm[l] + (size_t)t * slot; io.hmb_new[l] = sh ? m->dhm_b[l] + (size_t)t * slot : nullptr;
        io.drop_h[l] = drop_site(m, l + 2, (long long)t * (long long)slot);
      }
      io.drop_out = drop_site(m, 16, (long long)t * (long long)slot);
    }
    dec_step_forward(m, io, T);
  }
  if (chain_ctxa && keep_gates)                                    // q = W_a h_top of all L steps for the backward pass's d(context) (attention_dctx), as behind the whole-sequence kernel
    gemm_hh(s, m->dhs_b[m->Ld - 1] + slot, Hd, m->swa.wb, Hd, m->q_all, Hd, L * B, Hd, Hd, nullptr, nullptr, 0);
  }
  prof_mark(m, AOCR_PROF_RNN_GEMM);
  // (a training step without dropout: the projector runs in loss_and_dlogits' launch, with the criterion and its own data gradient)
  m->proj_fused = keep_gates && bf && !m->drop_on && project_loss_ok(L * B, m->V, Hd);
  if (!m->proj_fused) gemm(s, bf, m->out_all + slot, Hd, true, m->wo, Hd, true, m->logits, LOGIT_LD, L * B, m->V, Hd, m->bo, nullptr, 0);
}

void loss_and_dlogits(aocr_model* m, const Dims& d, const int32_t* tge, int64_t st, int64_t sb, float grad_scale, bool want_grad,
                      float* loss_dev) {
  const int64_t rows = (int64_t)d.L * d.B;
  prof_mark(m, AOCR_PROF_OTHER);
  // (round 5, measured and dropped: d logits and the projector's data gradient d out_proj = d logits W_o in this pass -- d logits kept in the wave, W_o (80 KB, fp32) in LDS,
  //  exact fp32 FMAs instead of the K = 39 bf16 product behind it: 9 + 27 us of launches became one of ~60 us: 39 x 8 dependent LDS reads per row and lane)
  m->dout_ready = false;
  if (m->proj_fused && want_grad) {
    prof_mark(m, AOCR_PROF_RNN_GEMM);
    project_loss(m->s, m->out_all + (size_t)d.B * m->Hd, m->Hd, m->wo, m->bo, m->logits, m->dlogits, LOGIT_LD, m->nll_rows, m->dout_proj, tge, st, sb, d.B, (int)rows, m->V, m->Hd, grad_scale);
    m->dout_ready = true;
  } else {
    if (m->proj_fused) gemm(m->s, m->bf16, m->out_all + (size_t)d.B * m->Hd, m->Hd, true, m->wo, m->Hd, true, m->logits, LOGIT_LD, (int)rows, m->V, m->Hd, m->bo, nullptr, 0);
    logsoftmax_nll(m->s, m->logits, LOGIT_LD, tge, st, sb, d.B, nullptr, want_grad ? m->dlogits : nullptr, m->nll_rows, rows, m->V, grad_scale);
  }
  m->proj_fused = false;
  // the step's loss (sum of the rows' NLL, fp64 in one workgroup: deterministic): nothing on the device reads it, so a training step sums it behind the decoder BPTT kernel
  // (decoder_backward) instead of between the two whole-sequence kernels (12 us of launch + latency on a mostly idle chip)
  if (loss_dev) { if (want_grad && !m->prof_on) m->loss_pending = loss_dev; else sum_to_scalar(m->s, m->nll_rows, rows, loss_dev); }
}

static bool side_create(aocr_model* m);
// the side stream of the backward pass (created on first use, lowest priority) -- AOCR_NO_SIDE_WGRAD=1 keeps everything on one stream
static bool side_stream_on(aocr_model* m, int B, int T) {
  if (m->prof_on || !m->bf16 || getenv("AOCR_NO_SIDE_WGRAD")) return false;
  // The encoder BPTT cluster kernel needs every workgroup of a group resident at once; beside it the side stream's GEMM workgroups
  // must find free compute units, or they delay the group members that are placed last (bounded spins, aocr_cluster_status).
  // C3: 16 groups x 2 directions x 4 members = 128 of 256 units -> on.  He = 512 at batch 400: 256 workgroups per pass -> off.
  {
    int G = 0, RT = 0, groups = 0;
    if (cluster_ok(m, B, T, G, RT, groups)) {
      static const int cus = [] { int n = 0, dev = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n; }();
      const int per_pass = std::max(8, (cus - comm_reserved_cus(m)) / (8 * G) * 8);
      if ((std::min(2 * groups, per_pass) * G + comm_reserved_cus(m)) * 4 > cus * 3) return false;
      // a stacked encoder's BPTT runs up to Le cluster launches at once on lay_s[] (encoder_backward_pipe), each needing its groups co-resident;
      // the occupancy estimate above is for ONE launch -- no side stream beside the wavefront
      if (layer_pipe_chunks(m, T, G, groups) > 0) return false;
    }
  }
  return side_create(m);
}
static bool side_create(aocr_model* m) {
  if (!m->side) {
    int lo = 0, hi = 0; hipDeviceGetStreamPriorityRange(&lo, &hi);            // lo = numerically greatest = lowest priority
    const hipError_t e = getenv("AOCR_SIDE_PLAIN") ? hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking) : hipStreamCreateWithPriority(&m->side, hipStreamNonBlocking, lo);
    if (e != hipSuccess) { m->side = nullptr; return false; }
    if (hipEventCreateWithFlags(&m->side_go, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&m->side_done, hipEventDisableTiming) != hipSuccess) return false;
    if (hipStreamCreateWithPriority(&m->side2, hipStreamNonBlocking, lo) != hipSuccess) m->side2 = nullptr;      // optional (decoder_backward)
    if (m->side2 && hipEventCreateWithFlags(&m->side2_done, hipEventDisableTiming) != hipSuccess) m->side2_done = nullptr;
  }
  return m->side_go && m->side_done;
}
// The per-token gate-input table of the first decoder layer (decoder_tf_forward) depends on the parameters only: a training step computes it
// on the side stream while the CNN runs (20 us off the main stream between the encoder and the decoder kernels at C3).
// Start of a training step (aocr_train_forward_backward): what depends on nothing but the parameters -- zeroing the gradient vector (49 MB), the
// bf16 weight shadows (100 MB of traffic), the token table -- goes to the side stream and runs under conv1 (VALU-bound, 40 us) instead of in front
// of it; cnn_forward waits for the shadows behind conv1, backward_all for the zeroed gradients.  AOCR_NO_SIDE_PROLOGUE=1: everything in line.
void step_prologue(aocr_model* m, size_t grad_bytes) {             // grad_bytes = 0: a decode call (no gradient vector to zero)
  m->tab_ready = m->zero_pending = m->shadow_pending = m->shadow2_pending = m->tab_valid = false;
  auto ev = [](hipEvent_t& e) { return e || hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
  // (bf16 mode only: in exact-fp32 mode the same move -- gradient zeroing and the 16 per-step weight transposes beside the forward pass -- measured SLOWER, C2 7.52 -> 7.74 ms:
  //  the forward pass there is a chain of ~150 small dependent launches, and the side stream's launches get in their way)
  const int bn_n = 2 * (256 + 512 + 512);
  const bool side = m->bf16 && !m->prof_on && !getenv("AOCR_NO_SIDE_WGRAD") && !env_on("AOCR_NO_SIDE_PROLOGUE") && side_create(m) && ev(m->zero_done) && ev(m->shadow_done) && ev(m->tab_done);
  if (!side) {
    if (grad_bytes) { hipMemsetAsync(m->grads, 0, grad_bytes, m->s); if (m->bn_snap) step_snapshot(m->s, m->bn_state, m->bn_snap, bn_n, m->cl_err); }
    return;
  }
  hipEventRecord(m->side_go, m->s); hipStreamWaitEvent(m->side, m->side_go, 0);       // behind whatever wrote the parameters on the model's stream
  if (grad_bytes && m->bn_snap && m->shadow_host.empty()) step_snapshot(m->s, m->bn_state, m->bn_snap, bn_n, m->cl_err);
  if (!m->shadow_host.empty()) {
    // conv2's taps first, with an event of their own: conv2 starts as soon as conv1 is done (the rest of the table -- 100 MB of traffic -- runs under conv2
    // and is joined in front of conv3): 25 us off the head of the step at C3, where conv2 waited for the whole table and the cross-stream hand-over
    const bool split = m->shadow_tiles_conv2 > 0 && ev(m->shadow2_done) && !env_on("AOCR_NO_SHADOW_SPLIT");
    if (split) {
      shadow_jobs(m->side, m->shadow_dev, (int)m->shadow_host.size(), m->shadow_tiles_conv2, 0);
      hipEventRecord(m->shadow2_done, m->side);
      shadow_jobs(m->side, m->shadow_dev, (int)m->shadow_host.size(), m->shadow_tiles - m->shadow_tiles_conv2, m->shadow_tiles_conv2);
    } else shadow_jobs(m->side, m->shadow_dev, (int)m->shadow_host.size(), m->shadow_tiles);
    // snapshot of the running statistics (10 KB; joined with the shadows in front of the first BatchNorm layer, behind conv2's taps so that it delays nothing)
    if (grad_bytes && m->bn_snap) step_snapshot(m->side, m->bn_state, m->bn_snap, bn_n, m->cl_err);
    hipEventRecord(m->shadow_done, m->side); m->shadow_pending = true; m->shadow2_pending = split;
  }
  if (grad_bytes) { hipMemsetAsync(m->grads, 0, grad_bytes, m->side); hipEventRecord(m->zero_done, m->side); m->zero_pending = true; }
  if (m->bzx_tab && !env_on("AOCR_NO_EMB_TABLE") && segsum_supported(4 * m->Hd, m->V, m->E)) {
    const LstmP& p1 = m->dec[0];
    gemm(m->side, true, m->lookup, m->E, true, p1.wi, p1.in, true, m->bzx_tab, 4 * m->Hd, m->V, 4 * m->Hd, m->E, p1.bi, p1.bh, 0);
    hipEventRecord(m->tab_done, m->side);
    m->tab_ready = true;
  }
}

// decoder BPTT, model.lua:643-661, t = L..1.
static void decoder_backward(aocr_model* m, const Dims& d, const int32_t* tgt) {
  hipStream_t s = m->s; const bool bf = m->bf16;
  const int B = d.B, T = d.T, L = d.L, Hd = m->Hd, E = m->E, V = m->V, Ld = m->Ld;
  const size_t slot = (size_t)B * Hd;
  const int rows = L * B;
  // projector backward for all steps at once (model.lua:648): d(out) part, gradWeight, gradBias
  prof_mark(m, AOCR_PROF_RNN_GEMM);
  if (!m->dout_ready) gemm(s, bf, m->dlogits, LOGIT_LD, true, m->wo, Hd, false, m->dout_proj, Hd, rows, Hd, V, nullptr, nullptr, 0);
  m->dout_ready = false;
  // (gradWeight of the projector: nothing in this pass reads it -- with the hoisted parameter gradients below, beside the encoder BPTT: 41 us off the main stream at C3)
  ColsumJobs cj; cj.n = 0; cj.total = 0;                          // projector bias + the LSTM biases of every layer: one launch at the end of this pass
  colsum_defer(cj, m->dlogits, LOGIT_LD, rows, V, m->dbo);
  prof_mark(m, AOCR_PROF_DEC_BWD);
  const bool cl_bwd = m->dgates_il && dec_cluster_bwd_ok(m, T, L);
  // (the launch chain accumulates into the initial-state gradients; the whole-sequence kernel writes every element of them at its end -- no zeroing launch in front of it)
  if (!cl_bwd) { ZeroList zl; for (int l = 0; l < Ld; ++l) { zl.add(m->dh_rec[l], slot * sizeof(float)); zl.add(m->dc_st[l], slot * sizeof(float)); } zero_many(s, zl); }
  const bool feed_fused = m->cfg.input_feed && Ld <= 2 && !m->drop_on;  // the feed product joins the grouped launch and carries the tanh backward (dropout: the separate d pre kernel knows the mask)
  if (!cl_bwd && m->q_pending) { hipStreamWaitEvent(s, m->q_done, 0); m->q_pending = false; }      // the launch chain's attention backward reads q step by step
  if (cl_bwd) {     // the whole loop as one launch (dec_cluster.hip); needs the forward cluster kernel's saved state
    DecClBwdArgs a; a.B = B; a.T = T; a.L = L; a.epoch = next_epoch(m);
    a.w2i_t = m->dec[1].swi.wtb; a.w2h_t = m->dec[1].swh.wtb; a.w1h_t = m->dec[0].swh.wtb; a.w1f_t = m->dec[0].swi.wtb;
    a.wc_t = m->swc.wtb; a.wa_t = m->swa.wtb; a.dout_proj = m->dout_proj; a.out = m->out_all; a.a_all = m->a_all; a.ctxb = m->context_b;
    for (int l = 0; l < 2; ++l) { a.cs[l] = m->dcs[l]; a.gates[l] = m->dgates[l]; a.dz[l] = m->ddz[l]; a.dzb[l] = m->ddz_b[l]; a.dc_st[l] = m->dc_st[l]; a.dh_rec[l] = m->dh_rec[l]; }
    a.dpre = m->dpre_all; a.dpre_b = m->dpre_b; a.dcat = m->dcat_all; a.ds_all = m->ds_all; a.dq = m->dq_all; a.dq_b = m->dq_b; a.dfeed = m->dfeed;
    a.xbuf = m->dc_bxbuf; a.xtab = m->dc_xtab; a.err = m->cl_err;
    if (m->drop_on) { a.drop_h = drop_site(m, 2, 0); a.drop_out = drop_site(m, 16, 0); }      // the forward cluster kernel ran with these masks
    dec_cluster_backward(s, a);
  } else
  for (int t = L - 1; t >= 0; --t) {
    const bool last = t == L - 1;
    const float* out_t = m->out_all + (size_t)(t + 1) * slot;
    float* dpre = m->dpre_all + (size_t)t * slot;
    float* dcat = m->dcat_all + (size_t)t * B * 2 * Hd;
    // d(tanh) with the input-feed gradient of step t+1 added (model.lua:649,654-657).  With input feed the product
    // d(prev attention output) of step t+1 already wrote dpre(t) through its epilogue (see the grouped launch below).
    const bool sh = m->bf16 && m->dpre_b != nullptr;
    if (!m->cfg.input_feed || last || !feed_fused)
      { const DropSpec dsp = drop_site(m, 16, (long long)t * (long long)slot);
        dpre_tanh(s, m->dout_proj + (size_t)t * slot, (m->cfg.input_feed && !last) ? m->dfeed : nullptr, out_t, dpre, (int64_t)slot,
                  sh ? m->dpre_b + (size_t)t * slot : nullptr, &dsp); }
    LoadKh2 dpa = make_loadkh(sh ? m->dpre_b + (size_t)t * slot : nullptr, Hd, B, Hd);
    run_store_nn(m, make_loadk(dpre, Hd, B, Hd), m->swc, make_store(dcat, 2 * Hd, B, 2 * Hd), B, &dpa);      // d[c ; h_top] = dpre W_c
    const bool bwd_ctxa = sh && m->ctxa_b && attention_dual_ok(T, Hd, m->context_b, m->ctxa_b);     // (the forward chain of this step made the same choice: same conditions)
    if (bwd_ctxa)       // d q as before (d W_a, hoisted) AND d h_top's attention part = sum_t d s[t] (ctx W_a)[t], into the decode path's idle q buffer
      attention_backward_dual(s, m->a_all + (size_t)t * B * T, dcat, 2 * Hd, m->ds_all + (size_t)t * B * T, m->dq_all + (size_t)t * slot, m->dq_b + (size_t)t * slot, m->bq, B, T, m->context_b, m->ctxa_b);
    else
    attention_backward(s, m->context, m->q_all + (size_t)t * slot, m->a_all + (size_t)t * B * T, dcat, 2 * Hd,
                       m->ds_all + (size_t)t * B * T, m->dq_all + (size_t)t * slot, B, T, Hd, sh ? m->dq_b + (size_t)t * slot : nullptr, m->context_b,
                       m->cat_all + (size_t)t * B * 2 * Hd, 2 * Hd);      // (the forward pass's weighted context: the streamed kernel's one-pass form)
    // top layer: d h_top = dq W_a + dcat[:, Hd:] + recurrent part
    for (int l = Ld - 1; l >= 0; --l) {
      LoadK la; LoadKh2 lah; EpGatesBwd e; const ShW* ww;
      if (l == Ld - 1 && bwd_ctxa) {                                  // no product: d h_top = attention part + d cat[:, Hd:] + recurrent part
        la = make_loadk(m->dq_all + (size_t)t * slot, Hd, B, 0); ww = &m->swa;
        lah = make_loadkh(nullptr, Hd, B, 0);
        e.dh1 = dcat + Hd; e.ld1 = 2 * Hd; e.dh3 = m->bq; e.ld3 = Hd;
      } else if (l == Ld - 1) {
        la = make_loadk(m->dq_all + (size_t)t * slot, Hd, B, Hd); ww = &m->swa;
        lah = make_loadkh(sh ? m->dq_b + (size_t)t * slot : nullptr, Hd, B, Hd);
        e.dh1 = dcat + Hd; e.ld1 = 2 * Hd;
      } else {                                                        // from the layer above: dz_{l+1} W_{l+1,i2h}
        la = make_loadk(m->ddz[l + 1] + (size_t)t * B * 4 * Hd, 4 * Hd, B, 4 * Hd); ww = &m->dec[l + 1].swi;
        lah = make_loadkh(sh ? m->ddz_b[l + 1] + (size_t)t * B * 4 * Hd : nullptr, 4 * Hd, B, 4 * Hd);
        e.dh1 = nullptr; e.ld1 = 0;
      }
      e.dh2 = m->dh_rec[l]; e.ld2 = Hd;
      e.dc_in = m->dc_st[l]; e.lddc = Hd;
      e.gates = m->dgates[l] + (size_t)t * B * 4 * Hd; e.ldg = 4 * Hd; e.gil = m->dgates_il;
      if (l < Ld - 1) e.drop = drop_site(m, l + 2, (long long)t * (long long)slot);            // the layer above read Dropout(h_l)
      e.c_prev = m->dcs[l] + (size_t)t * slot; e.ldcp = Hd; e.c = m->dcs[l] + (size_t)(t + 1) * slot; e.ldcc = Hd;
      e.dz = m->ddz[l] + (size_t)t * B * 4 * Hd; e.lddz = 4 * Hd; e.dc_out = m->dc_st[l]; e.lddco = Hd; e.M = B; e.H = Hd;
      if (sh) { e.dzb = m->ddz_b[l] + (size_t)t * B * 4 * Hd; e.lddzb = 4 * Hd; }
      run_gates_bwd(m, 1, &la, &ww, &e, B, Hd, &lah);
    }
    // Products of this step's d z that are only consumed at step t-1 -- the recurrent parts dz_l W_{l,h2h} and the
    // input-feed gradient d(prev attention output) = dz_1 W_{1,i2h}[:, E:] -- are independent: one grouped launch.
    {
      LoadK ga[3]; LoadKh2 gah[3]; const ShW* gw[3]; EpStore gep[3]; int n = 0;
      for (int l = Ld - 1; l >= 0 && n < 3; --l) {
        const size_t zo = (size_t)t * B * 4 * Hd;
        ga[n] = make_loadk(m->ddz[l] + zo, 4 * Hd, B, 4 * Hd); gah[n] = make_loadkh(sh ? m->ddz_b[l] + zo : nullptr, 4 * Hd, B, 4 * Hd);
        gw[n] = &m->dec[l].swh; gep[n] = make_store(m->dh_rec[l], Hd, B, Hd); ++n;
      }
      const bool feed_grouped = m->cfg.input_feed && n < 3 && Ld <= 2 && !m->drop_on;
      if (feed_grouped) {
        const size_t zo = (size_t)t * B * 4 * Hd;
        ga[n] = make_loadk(m->ddz[0] + zo, 4 * Hd, B, 4 * Hd); gah[n] = make_loadkh(sh ? m->ddz_b[0] + zo : nullptr, 4 * Hd, B, 4 * Hd);
        gw[n] = &m->dec[0].swi;
        if (t > 0) {                                      // writes dpre(t-1) = (dout_proj(t-1) + this product) * (1 - out(t-1)^2) directly
          gep[n] = make_store(m->dpre_all + (size_t)(t - 1) * slot, Hd, B, Hd);
          gep[n].dg = m->dout_proj + (size_t)(t - 1) * slot; gep[n].dout = m->out_all + (size_t)t * slot; gep[n].ldd = Hd;
          if (sh) { gep[n].Cb = m->dpre_b + (size_t)(t - 1) * slot; gep[n].ldcb = Hd; }
        } else gep[n] = make_store(m->dfeed, Hd, B, Hd);  // step 0: nothing consumes it
        ++n;
      }
      if (Ld <= 3) run_store_nn_group(m, n, ga, gw, gep, B, gah);
      else for (int l = Ld - 1; l >= 0; --l) {           // more layers than one launch groups: one launch each
        const size_t zo = (size_t)t * B * 4 * Hd;
        LoadKh2 dza = make_loadkh(sh ? m->ddz_b[l] + zo : nullptr, 4 * Hd, B, 4 * Hd);
        run_store_nn(m, make_loadk(m->ddz[l] + zo, 4 * Hd, B, 4 * Hd), m->dec[l].swh, make_store(m->dh_rec[l], Hd, B, Hd), B, &dza);
      }
      if (m->cfg.input_feed && !feed_grouped) {
        LoadKh2 dz0 = make_loadkh(sh ? m->ddz_b[0] + (size_t)t * B * 4 * Hd : nullptr, 4 * Hd, B, 4 * Hd);
        run_store_nn(m, make_loadk(m->ddz[0] + (size_t)t * B * 4 * Hd, 4 * Hd, B, 4 * Hd), m->dec[0].swi, make_store(m->dfeed, Hd, B, Hd), B, &dz0);
      }
    }
  }
  // d(context), model.lua:652-653 summed over the loop: the ONE result of this pass the encoder BPTT waits for
  prof_mark(m, AOCR_PROF_RNN_GEMM);
  // The hoisted parameter gradients below need the BPTT's outputs only: their streams are released HERE, in front of the loss sum and d(context) (67 us
  // during which the chip is mostly idle), not behind them -- the side streams were the critical path of this section (the main stream idled ~70 us at
  // the join in front of the CNN backward pass).  The first side-stream kernels are small (token sums, the projector's 96-workgroup product): the encoder
  // BPTT kernel still finds its compute units when it starts.  AOCR_SIDE_GO_LATE=1: the old place.
  const bool side_on = side_stream_on(m, B, T), side_early = side_on && !env_on("AOCR_SIDE_GO_LATE");
  if (side_early) hipEventRecord(m->side_go, s);
  if (m->loss_pending && !side_early) { sum_to_scalar(s, m->nll_rows, (int64_t)L * B, m->loss_pending); m->loss_pending = nullptr; }      // the step's loss (loss_and_dlogits): behind the BPTT, in front of every event the exchange waits for (side streams released early: on the side stream below, off the chain the encoder BPTT waits for)
  if (m->q_pending) { hipStreamWaitEvent(s, m->q_done, 0); m->q_pending = false; }          // q of all steps (decoder_tf_forward put the product on the side stream)
  attention_dctx(s, m->a_all, m->ds_all, m->dcat_all, 2 * Hd, m->q_all, m->dctx, L, B, T, Hd);
  // ---- hoisted parameter gradients (accGradParameters of every clone summed over time).  Nothing downstream of them but the
  // optimizer: they run on the side stream beside the encoder BPTT, whose whole-sequence kernel occupies HALF the compute units
  // (16 groups x 2 directions x 4 members at C3) for ~0.2 ms.  The side stream has the lowest priority, so the encoder kernel's
  // workgroups are placed first.  Off while the per-family profile marks are on (the marks live on the model's stream).
  hipStream_t ms = s;
  if (side_on) {
    if (!side_early) hipEventRecord(m->side_go, ms);
    s = m->side; hipStreamWaitEvent(s, m->side_go, 0); m->side_busy = true;
    if (m->loss_pending) { sum_to_scalar(s, m->nll_rows, (int64_t)L * B, m->loss_pending); m->loss_pending = nullptr; }
  }
  // round 4: a SECOND side stream for the latency- / HBM-bound part of this section (projector gradWeight: a 24-way split-K product of 96 workgroups;
  // the bias column sums: 67 us over 180 MB) so that it runs beside the weight-gradient GEMMs instead of in front of them -- with the encoder's weight
  // gradients on the side stream too, that stream had become the critical path (the main stream idled ~120 us at the join).  It rejoins the first side
  // stream at the end of this function, so "the side stream is done" still means "every hoisted gradient is done".
  hipStream_t s2 = s;
  if (m->side_busy && m->side2 && m->side2_done && !env_on("AOCR_NO_SIDE2")) { s2 = m->side2; hipStreamWaitEvent(s2, m->side_go, 0); }
  gemm(s2, bf, m->dlogits, LOGIT_LD, false, m->out_all + slot, Hd, false, m->dwo, Hd, V, Hd, rows, nullptr, nullptr, EP_ATOMIC);
  const float* h_top_all = m->dhs[Ld - 1] + slot;
  WGradProblem wg[16]; int nwg = 0;
  const bool sh = m->bf16 && m->dpre_b != nullptr;
  wg[nwg++] = WGradProblem{m->dpre_all, Hd, m->cat_all, 2 * Hd, m->dwc, 2 * Hd, Hd, 2 * Hd, rows, m->dpre_b, m->cat_b};
  wg[nwg++] = WGradProblem{m->dq_all, Hd, h_top_all, Hd, m->dwa, Hd, Hd, Hd, rows, m->dq_b, sh ? m->dhs_b[Ld - 1] + slot : nullptr};
  for (int l = 0; l < Ld; ++l) {
    const LstmP& p = m->dec[l]; const float* dz = m->ddz[l];
    const bf16_t* dzb = sh ? m->ddz_b[l] : nullptr;
    wg[nwg++] = WGradProblem{dz, 4 * Hd, m->dhs[l], Hd, p.dwh, Hd, 4 * Hd, Hd, rows, dzb, sh ? m->dhs_b[l] : nullptr};
    if (cj.n >= 8) colsum_flush(s2, cj);
    if (l == 0 && m->emb_table) {
      // the embedding side through the sums of d z by token (ops_misc.hip: segsum_by_token): S [V][4 Hd] in one pass over d z, which also yields
      // the layer's bias gradients; then d lookup += S W_i2h[:, :E] and d W_i2h[:, :E] = S^T lookup -- two V-sized products in exact fp32
      // (on the FIRST side stream, in front of the weight-gradient GEMMs: with the sums on the second stream the GEMM's 228 workgroups were dispatched ahead of the
      //  encoder BPTT kernel's groups and delayed them -- enc_cl_bwd 246 -> 314 us, step 5.28 -> 5.32 ms)
      segsum_by_token(s, dz, 4 * Hd, tgt, 1, L, L, B, 4 * Hd, V, m->emb_seg, m->emb_index, p.dbi, p.dbh, p.wi, p.in, m->lookup, E, m->dlookup, p.dwi);
      if (m->cfg.input_feed) wg[nwg++] = WGradProblem{dz, 4 * Hd, m->out_all, Hd, p.dwi + E, p.in, 4 * Hd, Hd, rows, dzb, sh ? m->out_b : nullptr};
      continue;
    }
    colsum_defer(cj, dz, 4 * Hd, rows, 4 * Hd, p.dbi, p.dbh);
    if (l == 0) {
      wg[nwg++] = WGradProblem{dz, 4 * Hd, m->emb_all, E, p.dwi, p.in, 4 * Hd, E, rows};
      if (m->cfg.input_feed) wg[nwg++] = WGradProblem{dz, 4 * Hd, m->out_all, Hd, p.dwi + E, p.in, 4 * Hd, Hd, rows, dzb, sh ? m->out_b : nullptr};
      // N = E = 20 columns only: 48 row tiles would each walk all of K = 4Hd; split K over atomics instead (185 -> ~30 us at C3)
      hipMemsetAsync(m->demb_all, 0, (size_t)rows * E * sizeof(float), s);
      gemm(s, bf, dz, 4 * Hd, true, p.wi, p.in, false, m->demb_all, E, rows, E, 4 * Hd, nullptr, nullptr, EP_ATOMIC);
      embedding_scatter_accum(s, m->demb_all, tgt, 1, L, m->dlookup, L, B, E, V);
    } else {
      if (m->drop_on) wg[nwg++] = WGradProblem{dz, 4 * Hd, m->dhm[l - 1], Hd, p.dwi, Hd, 4 * Hd, Hd, rows, dzb, sh ? m->dhm_b[l - 1] : nullptr};      // the layer saw Dropout(h)
      else wg[nwg++] = WGradProblem{dz, 4 * Hd, m->dhs[l - 1] + slot, Hd, p.dwi, Hd, 4 * Hd, Hd, rows, dzb, sh ? m->dhs_b[l - 1] + slot : nullptr};
    }
  }
  colsum_flush(s2, cj);
  grouped_wgrad(s, bf, wg, nwg, m->wg_part, m->wg_part_floats);
  if (s2 != s) { hipEventRecord(m->side2_done, s2); hipStreamWaitEvent(s, m->side2_done, 0); }
}

// The flat gradient vector completes back to front: decoder + projector groups, then both encoder groups, then the CNN from conv7
// down.  An event marks each point so that a data-parallel caller can start summing a bucket while the rest of the backward pass
// still runs (aocr_grad_buckets / aocr_stream_wait_grads).
void backward_all(aocr_model* m, const float* images, const int32_t* tgt, const Dims& d) {
  m->side_busy = false;
  if (m->zero_pending) { hipStreamWaitEvent(m->s, m->zero_done, 0); m->zero_pending = false; }          // step_prologue zeroed the gradient vector on the side stream
  decoder_backward(m, d, tgt);
  // Exchange policy: NO collective is in flight while a whole-sequence kernel runs.  A collective's kernel stays
  // resident until every peer has joined it; a cluster kernel needs all members of a group resident at once and bounds its spins.  With
  // a communicator attached, bucket 0 (ready here) is therefore released only behind the encoder BPTT kernel -- it still has the whole
  // CNN backward pass to hide in.  AOCR_COMM_EARLY_BUCKET0=1 restores the early release; the encoder kernels then leave
  // comm_reserved_cus() compute units free (rnn_cluster.hip).
  const bool hold0 = comm_holds_bucket0(m);
  if (!hold0) hipEventRecord(m->grad_ev[0], m->side_busy ? m->side : m->s);          // decoder + projector gradients complete (on the side stream when it ran them)
  encoder_backward(m, d);
  // Round 6: no join in front of the CNN backward pass when its filter gradients run on the side stream anyway.  The encoder's weight gradients (a 96 us grouped
  // product + its slab sums at C3) start only when the BPTT kernel ends, so the main stream -- 50 us of d X -- idled ~90 us at the join; now BatchNorm 7's
  // backward pass and conv7's data gradient run beside them, the side stream's first filter gradient queues behind them (it shares their slab scratch: same
  // stream, in order), and cnn_backward's own join at its end covers both.  AOCR_JOIN_BEFORE_CNN_BWD=1: the join.
  const bool through = m->side_busy && cnn_wgrad_on_side(m) && !env_on("AOCR_JOIN_BEFORE_CNN_BWD");
  if (through) {                                                // the bucket events: behind both streams, recorded on the side stream
    hipEventRecord(m->cw_main, m->s); hipStreamWaitEvent(m->side, m->cw_main, 0);
    if (hold0) hipEventRecord(m->grad_ev[0], m->side);
    hipEventRecord(m->grad_ev[1], m->side);
  } else {
    if (m->side_busy) { hipEventRecord(m->side_done, m->side); hipStreamWaitEvent(m->s, m->side_done, 0); }   // join before the CNN backward fills the chip
    if (hold0) hipEventRecord(m->grad_ev[0], m->s);
    hipEventRecord(m->grad_ev[1], m->s);
  }
  cnn_backward(m, images, d);
  hipEventRecord(m->grad_ev[3], m->s);
}

// ------------------------------------------------------------------------------------------------
// beam search, model.lua:360-536 + back-trace :573-585.  Rows r = b*k + beam; the context is not replicated
// (the attention kernel maps row -> image with ctx_div).
// ------------------------------------------------------------------------------------------------
void decode_beam(aocr_model* m, const Dims& d, const int32_t* tgt, int beam, int32_t* labels, float* scores, const aocr_trie* trie) {
  hipStream_t s = m->s; const bool bf = m->bf16;
  const int B = d.B, T = d.T, Lt = d.L, Hd = m->Hd, E = m->E, V = m->V, Ld = m->Ld;
  const int k = beam;
  const LstmP& p1 = m->dec[0];
  float* c0[MAXL]; float* h0[MAXL];
  for (int l = 0; l < Ld; ++l) { c0[l] = m->bc[0][l]; h0[l] = m->bh[0][l]; }
  // The embedding part of the first layer's gate input depends on the token only: one table row per vocabulary entry
  // (lookup W_i2h[:, :E]^T + both biases, LSTM.lua:55-56,79-80), gathered per step instead of a K = 20 GEMM per step.
  if (m->tab_ready) { hipStreamWaitEvent(s, m->tab_done, 0); m->tab_ready = false; m->tab_valid = true; }       // step_prologue of this decode call computed it on the side stream
  if (!m->tab_valid) { gemm(s, bf, m->lookup, E, true, p1.wi, p1.in, true, m->bzx_tab, 4 * Hd, V, 4 * Hd, E, p1.bi, p1.bh, 0); m->tab_valid = true; }
  if (k == 1 && V <= 40 && m->out_b && m->dc_pbuf && !getenv("AOCR_NO_DEC_GREEDY") && dec_cluster_ok(m, T, Lt)) {
    // greedy decode: the whole loop (cell, attention, projector, LogSoftMax, selection) as one launch of the decoder cluster kernel
    float* tc0[MAXL]; float* th0[MAXL];
    for (int l = 0; l < Ld; ++l) { tc0[l] = m->dcs[l]; th0[l] = m->dhs[l]; }
    dec_init_state(m, d, tc0, th0, m->out_all, B, true);
    const size_t slot = (size_t)B * Hd; (void)slot;
    gemm_hh_shadow(s, m->context_b, Hd, m->swa.wtb, Hd, m->dctx, Hd, m->ctxa_b, Hd, B * T, Hd, Hd);
    m->ctxa_fresh = true;                                            // the gold pass of this call scores against the same context
    DecClFwdArgs a; a.B = B; a.T = T; a.L = Lt; a.epoch = next_epoch(m);
    a.w1i = m->dec[0].swi.wb; a.w1h = m->dec[0].swh.wb; a.w2i = m->dec[1].swi.wb; a.w2h = m->dec[1].swh.wb; a.wc = m->swc.wb;
    a.b2i = m->dec[1].bi; a.b2h = m->dec[1].bh; a.zx1 = m->bzx_tab; a.ctxb = m->context_b; a.ctxa = m->ctxa_b;
    for (int l = 0; l < 2; ++l) { a.cs[l] = m->dcs[l]; a.hsb[l] = m->dhs_b[l]; a.gates[l] = nullptr; }
    a.a_all = m->a_all; a.out = m->out_all; a.cat_b = m->cat_b; a.out_b = m->out_b;
    a.xbuf = m->dc_xbuf; a.xtab = m->dc_xtab; a.err = m->cl_err;
    a.tok0 = tgt; a.tok0_stride = Lt; a.wo = m->wo; a.bo = m->bo; a.V = V; a.pbuf = m->dc_pbuf; a.labels = labels; a.scores = scores;
    if (trie) { a.trie_mask = (const unsigned long long*)trie->child_mask_dev; a.trie_base = trie->child_base_dev; a.trie_child = trie->child_dev; }
    dec_cluster_forward(s, a, true);
    return;
  }
  if (k > 1 && m->out_b && m->dc_pbuf && dec_cluster_ok(m, T, Lt) && dec_chain_beam_supported(B, Lt, k, V)) {
    // beam search inside the decoder chain kernel: the k hypotheses of an image are rows of one chain; state gather by parent, LogSoftMax and the
    // k-best selection happen in the kernel (dec_chain.hip, BEAM), the back-trace is the launch chain's
    float* tc0[MAXL]; float* th0[MAXL];
    for (int l = 0; l < Ld; ++l) { tc0[l] = m->dcs[l]; th0[l] = m->dhs[l]; }
    dec_init_state(m, d, tc0, th0, m->out_all, B, true);
    gemm_hh_shadow(s, m->context_b, Hd, m->swa.wtb, Hd, m->dctx, Hd, m->ctxa_b, Hd, B * T, Hd, Hd);
    m->ctxa_fresh = true;
    const int npass = dec_chain_beam_passes(B, Lt, k);
    reserve_epochs(m, (unsigned)npass + 1);
    DecClFwdArgs a; a.B = B; a.T = T; a.L = Lt; a.epoch = next_epoch(m);
    for (int i = 1; i < npass; ++i) (void)next_epoch(m);
    a.w1i = m->dec[0].swi.wb; a.w1h = m->dec[0].swh.wb; a.w2i = m->dec[1].swi.wb; a.w2h = m->dec[1].swh.wb; a.wc = m->swc.wb;
    a.b2i = m->dec[1].bi; a.b2h = m->dec[1].bh; a.zx1 = m->bzx_tab; a.ctxb = m->context_b; a.ctxa = m->ctxa_b;
    for (int l = 0; l < 2; ++l) { a.cs[l] = m->dcs[l]; a.hsb[l] = m->dhs_b[l]; a.gates[l] = nullptr; }
    a.a_all = m->a_all; a.out = m->out_all; a.cat_b = m->cat_b; a.out_b = m->out_b;
    a.xbuf = m->dc_xbuf; a.xtab = m->dc_xtab; a.err = m->cl_err;
    a.tok0 = tgt; a.tok0_stride = Lt; a.wo = m->wo; a.bo = m->bo; a.V = V; a.pbuf = m->dc_pbuf; a.labels = labels; a.scores = scores;
    a.beam = k; a.hist_tok = m->hist_tok; a.hist_par = m->hist_par; a.beam_scores = m->beam_scores;
    if (trie) { a.trie_mask = (const unsigned long long*)trie->child_mask_dev; a.trie_base = trie->child_base_dev; a.trie_child = trie->child_dev; }
    dec_chain_beam_forward(s, a);
    beam_backtrace(s, m->hist_tok, m->hist_par, m->beam_scores, labels, scores, Lt, B, k);
    return;
  }
  // Round 6: in bf16 mode the chain keeps bf16 shadows of its beam state (the gate / attention epilogues write them, the gather by parent converts), so its step
  // products read both operands from shadows -- the kernels of the training chain (from ~130 rows stepl.h; at Hd = 1024 and T <= 64 scores against ctx W_a) instead
  // of the fp32-activation forms.  The products round the same fp32 values to bf16 either way.  AOCR_NO_DECODE_SHADOWS=1: off
  const bool dsh = m->bf16 && m->bcat_b && !env_on("AOCR_NO_DECODE_SHADOWS");
  const bool dctxa = dsh && m->ctxa_b && m->swa.wtb && attention_dual_ok(T, Hd, m->context_b, m->ctxa_b);
  if (dctxa) { if (!m->ctxa_fresh) gemm_hh_shadow(s, m->context_b, Hd, m->swa.wtb, Hd, m->dctx, Hd, m->ctxa_b, Hd, B * T, Hd, Hd); m->ctxa_fresh = true; }      // (the gold pass of this call scores against the same context)
  dec_init_state(m, d, c0, h0, m->bfeed[0], B, false, dsh ? m->bh_b[0] : nullptr, dsh ? m->bfeed_b[0] : nullptr);      // the launch chain's beam buffers (the greedy cluster kernel above has its own: not initialised for nothing)
  int cur = 0;
  for (int t = 0; t < Lt; ++t) {
    const int kin = t == 0 ? 1 : k, R = B * kin;
    // current input tokens (t = 0: the GO column of the targets, model.lua:388; later: the tokens chosen one step ago)
    const int32_t* tok = t == 0 ? tgt : m->hist_tok + (size_t)(t - 1) * B * k;
    const int nxt = cur ^ 1;
    const bool direct = k == 1;                         // greedy: the parent of every row is itself, no state gather needed
    DecStepIO io; io.R = R; io.ctx_div = kin; io.feed = m->bfeed[cur];
    io.zx1 = m->bzx_tab; io.zx_tok = tok; io.zx_tok_stride = t == 0 ? Lt : 1;     // the gate epilogue gathers the token's table row
    for (int l = 0; l < Ld; ++l) {
      io.c_prev[l] = m->bc[cur][l]; io.h_prev[l] = m->bh[cur][l];
      io.c_new[l] = direct ? m->bc[nxt][l] : m->bc_new[l]; io.h_new[l] = direct ? m->bh[nxt][l] : m->bh_new[l];
      io.gates[l] = nullptr;
    }
    float* out = (direct && m->cfg.input_feed) ? m->bfeed[nxt] : m->bout;
    io.q = m->bq; io.a = m->ba; io.cat = m->bcat; io.out = out;
    if (dsh) {
      io.feed_b = m->bfeed_b[cur]; io.cat_b = m->bcat_b;
      io.out_b = (direct && m->cfg.input_feed) ? m->bfeed_b[nxt] : nullptr;       // (beam > 1: the gather below writes the next feed's shadow from the fp32 rows)
      for (int l = 0; l < Ld; ++l) { io.hb_prev[l] = m->bh_b[cur][l]; io.hb_new[l] = direct ? m->bh_b[nxt][l] : m->bh_new_b[l]; }
      if (dctxa) io.ctxa = m->ctxa_b;
    }
    dec_step_forward(m, io, T);
    // dictionary constraint: the node of every beam ping-pongs between trie_loc[0/1] (model.lua:380-387: all beams start at trie[2])
    TrieView tvs{}; const TrieView* tv = nullptr;
    if (trie) {
      tvs = TrieView{(const unsigned long long*)trie->child_mask_dev, trie->child_base_dev, trie->child_dev, m->trie_loc[t & 1],
                     m->trie_loc[(t & 1) ^ 1]};
      tv = &tvs;
    }
    if (V <= 64 && Hd % 4 == 0) {                        // projector + LogSoftMax + selection in one launch
      project_select(s, out, Hd, m->wo, m->bo, Hd, t == 0 ? nullptr : tok, m->beam_scores, m->hist_tok + (size_t)t * B * k,
                     m->hist_par + (size_t)t * B * k, B, kin, k, V, tv);
    } else {
      SmallKKArgs z; z.a = make_loadk(out, Hd, R, Hd); z.b = make_loadk(m->wo, Hd, V, Hd);
      z.ep = make_store(m->blogits, LOGIT_LD, R, V, m->bo, nullptr, 0); z.K = Hd;
      launch_small_kk(s, bf, 1, &z, R, V);
      logsoftmax_nll(s, m->blogits, LOGIT_LD, tgt, 0, 0, R, m->blogp, nullptr, nullptr, R, V, 0.f);
      beam_select(s, m->blogp, t == 0 ? nullptr : tok, m->beam_scores, m->hist_tok + (size_t)t * B * k,
                  m->hist_par + (size_t)t * B * k, B, kin, k, V, nullptr, 0, tv);
    }
    if (!direct) {
      const int32_t* par = m->hist_par + (size_t)t * B * k;
      const float* gs[2 * MAXL + 1]; float* gd[2 * MAXL + 1]; bf16_t* gb[2 * MAXL + 1]; int ng = 0;      // model.lua:521-535: gather states by parent beam, one launch
      for (int l = 0; l < Ld; ++l) { gs[ng] = m->bc_new[l]; gb[ng] = nullptr; gd[ng++] = m->bc[nxt][l]; gs[ng] = m->bh_new[l]; gb[ng] = dsh ? m->bh_b[nxt][l] : nullptr; gd[ng++] = m->bh[nxt][l]; }
      if (m->cfg.input_feed) { gs[ng] = m->bout; gb[ng] = dsh ? m->bfeed_b[nxt] : nullptr; gd[ng++] = m->bfeed[nxt]; }
      if (ng <= 8) gather_beam_rows_many(s, ng, gs, gd, Hd, par, B, kin, k, Hd, dsh ? gb : nullptr);
      else for (int i = 0; i < ng; ++i) { gather_beam_rows(s, gs[i], Hd, gd[i], Hd, par, B, kin, k, Hd); if (dsh && gb[i]) copy2d_bf16(s, gd[i], Hd, gb[i], Hd, B * k, Hd); }
    }
    cur = nxt;
  }
  beam_backtrace(s, m->hist_tok, m->hist_par, m->beam_scores, labels, scores, Lt, B, k);
}

}  // namespace aocr
