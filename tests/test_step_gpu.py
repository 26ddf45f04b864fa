"""Fused-path parity: aocr.Model (HIP, through the C ABI) against the CPU float64 oracle on the same seeded
inputs and injected weights: CNN features, context, decoder logits (the 1e-4 target of BASELINE.json), loss,
every gradient tensor, the clipped SGD update, and greedy / beam decode + gold pass."""
import numpy as np
import pytest
import torch

from tol import check_logits

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4          # BASELINE.json north_star: decoder logits within 1e-4 max-abs (fp32 MFMA path)


def make(cfgkw, B, W, maxlen, compute="f32", seed=910820, max_decoder_l=12, max_beam=5):
    import aocr
    import oracle_torch as O
    ocfg = O.OcrConfig(**cfgkw)
    P = O.init_params(ocfg, seed)
    st = O.init_bn_state()
    img, tgt, tge, nnz = O.synth_batch(B, W, max_len=maxlen, min_len=min(2, maxlen))
    m = aocr.Model()
    m._set_structure(dict(encoder_num_hidden=ocfg.enc_hidden, encoder_num_layers=ocfg.enc_layers,
                          decoder_num_layers=ocfg.dec_layers, input_feed=ocfg.input_feed))
    m._set_runtime(dict(batch_size=B, max_img_w=W, max_decoder_l=max_decoder_l, max_beam=max_beam, compute=compute))
    m.optim_state = {"learningRate": 0.1}
    m._build()
    m.set_parameters(P, st)
    batch = [img, tgt, tge, nnz, [f"img{i}" for i in range(B)]]
    return m, O, ocfg, P, st, batch


def relerr(got, ref):
    got = got.double(); ref = ref.double()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-12)).item()


def cosine(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def assert_grads_agree_up_to_decisions(ga, gb, what):
    """Two kernel paths that sum the same products in a different order: every tensor above the CNN agrees to summation-order
    noise (bf16 re-rounding included); a CNN gradient additionally sees the few ReLU / arg-max decisions that a last-bit
    difference flips (each moves ONE term of a sum: percent-level in max-norm, nothing in direction -- DESIGN.md section 4,
    test_c2_gradient_residual_is_decision_flips)."""
    worst = ("", 0.0)
    for k in ga:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):      # bias in front of a BatchNorm: exact gradient 0, only rounding noise
            continue
        e, c = relerr(gb[k], ga[k]), cosine(gb[k], ga[k])
        if e > worst[1]: worst = (k, e)
        if k.startswith("cnn."):
            assert e < 0.3 and c > 0.995, (what, k, e, c)    # measured cosine 0.998-0.999 (max-norm up to 0.18 on single entries at batch 2-4), the level of GPU-vs-oracle WITHOUT imposed decisions (with them: 0.99998, test_c3_bf16_all_gradients_vs_oracle)
        else:
            assert e < 3e-2 and c > 0.9999, (what, k, e, c)
    print(f"[parity] {what}: worst gradient rel {worst[1]:.3e} ({worst[0]})")


CASES = [
    dict(enc_hidden=32, enc_layers=1, dec_layers=2, input_feed=True),
    dict(enc_hidden=32, enc_layers=1, dec_layers=2, input_feed=False),
    dict(enc_hidden=32, enc_layers=1, dec_layers=1, input_feed=True),
    dict(enc_hidden=48, enc_layers=2, dec_layers=3, input_feed=True),
    dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True),     # every K a multiple of 64: staged bf16 step kernel
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_train_step_parity_small(cuda, case):
    m, O, ocfg, P, st, batch = make(CASES[case], B=5, W=36, maxlen=6)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    loss_ref, G, aux, st_new = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    B = img.shape[0]
    print(f"[parity] case {case}: loss hip {loss:.6f} oracle {float(loss_ref) * B:.6f}")
    feats = m.get_tensor("feats").transpose(0, 1)               # (T,B,512) -> (B,T,512)
    e = (feats.double() - aux["feats"]).abs().max().item(); print(f"[parity] feats max-abs {e:.3e}"); assert e < 2e-4
    e = (m.get_tensor("context").double() - aux["context"]).abs().max().item(); print(f"[parity] context max-abs {e:.3e}"); assert e < 1e-4
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item(); print(f"[parity] logits max-abs {e:.3e}"); check_logits(lg, aux["logits"], "f32", f"small case {case}")
    assert abs(loss - float(loss_ref) * B) < 1e-3 * max(1.0, abs(float(loss_ref) * B))
    e = relerr(m.get_tensor("dcontext"), aux["dctx"]); print(f"[parity] dcontext rel {e:.3e}"); assert e < 1e-3
    e = relerr(m.get_tensor("dfeats").transpose(0, 1), aux["dfeats"]); print(f"[parity] dfeats rel {e:.3e}"); assert e < 1e-3
    grads = m.get_gradients()
    worst = 0.0
    for k, g in G.items():
        e = relerr(grads[k], g)
        if g.abs().max() < 1e-9:                         # conv biases in front of BatchNorm have ~0 gradient
            e = (grads[k].double() - g).abs().max().item()
        worst = max(worst, e)
        print(f"[parity] grad {k:22s} rel {e:.3e} (max {g.abs().max().item():.3e})")
        assert e < 2e-3, k
    # running statistics
    bn = m.get_bn_state()
    for k, v in st_new.items():
        assert (bn[k].double() - v).abs().max().item() < 1e-5, k
    # optimizer, optim_sgd.lua:38-95 (lr 0.1, clip 5 and a tight clip that is active)
    for clip in (5.0, 0.05):
        m.set_parameters(P, st)
        m.train_forward_backward(batch)
        norms = m.sgd_step(lr=0.1, clip=clip)
        newP, nref = O.sgd_list(P, G, 0.1, clip)
        for gi in range(5):
            assert abs(norms[gi, 0] - nref[gi][0]) < 1e-4 * max(1, nref[gi][0])
            assert abs(norms[gi, 1] - nref[gi][1]) < 2e-3 * max(1e-3, nref[gi][1]), (gi, norms[gi, 1], nref[gi][1])
        got = m.get_parameters()
        for k, v in newP.items():
            assert (got[k].double() - v).abs().max().item() < 2e-5, k
    m.shutdown()


@pytest.mark.parametrize("W,compute", [(70, "f32"), (132, "f32"), (70, "bf16"), (200, "bf16")])
def test_ragged_widths(cuda, W, compute):
    """Widths whose intermediate maps are odd (floor-mode pooling drops a column: cnn.lua:15,20) and a T > 64 case
    (generic attention kernel); batch not a multiple of the 32-row tile."""
    m, O, ocfg, P, st, batch = make(CASES[0], B=3, W=W, maxlen=5, compute=compute)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    print(f"[parity] W={W} {compute}: T={aux['context'].shape[1]} logits max-abs {e:.3e}; loss {loss:.4f} vs {float(loss_ref) * 3:.4f}")
    check_logits(lg, aux["logits"], compute, f"ragged W={W}")
    if compute == "f32":
        grads = m.get_gradients()
        for k in ("cnn.conv2.w", "cnn.conv4.w", "cnn.conv7.w", "enc_bw.l1.i2h.w", "dec.attn.wa", "dec.lookup"):
            r = relerr(grads[k], G[k]); print(f"[parity] W={W} grad {k} rel {r:.3e}"); assert r < 2e-3, k
    m.shutdown()


@pytest.mark.parametrize("case,beam", [(0, 1), (0, 5), (1, 3), (2, 1), (3, 5)])
def test_decode_parity_small(cuda, case, beam):
    m, O, ocfg, P, st, batch = make(CASES[case], B=4, W=36, maxlen=5, max_decoder_l=10)
    # make running stats non-trivial so eval-mode BatchNorm is exercised
    st = {k: (v + 0.05 if k.endswith("rm") else v * 1.3) for k, v in st.items()}
    m.set_parameters(P, st)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    ref = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=10)
    loss, stats = m.step(batch, True, beam)
    out = m._dec_out
    print(f"[parity] decode case {case} beam {beam}: loss {loss:.5f} vs {float(ref['loss']):.5f}; labels[0] {out.labels[0].tolist()}")
    assert np.array_equal(out.labels, ref["labels"].numpy().astype(np.int32))
    assert np.abs(out.scores - ref["scores"].numpy()).max() < 2e-3
    assert np.abs(out.gold_scores - ref["gold_scores"].numpy()).max() < 2e-3
    assert abs(loss - float(ref["loss"])) < 2e-3 * max(1.0, float(ref["loss"]))
    assert stats[1] == ref["num_correct"]
    m.shutdown()


def test_logits_c2_shape(cuda):
    """BASELINE config C2: 32x100, B=64, He=256, Ld=2, input feed; decoder logits within 1e-4 of the fp64 oracle."""
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=64, W=100, maxlen=23,
                                    max_decoder_l=24, max_beam=1)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    with torch.no_grad():
        r = O.forward_train(P, {k: v.clone() for k, v in st.items()}, ocfg, img, tgt, tge, training=True)
    logits, loss = m.forward_logits(batch, training=True)
    e = (logits.double() - r["logits"]).abs().max().item()
    print(f"[parity] C2 logits max-abs {e:.3e} (mean |logit| {r['logits'].abs().mean().item():.3f}, max {r['logits'].abs().max().item():.3f}); loss {loss:.4f} vs {float(r['loss']) * 64:.4f}")
    check_logits(logits, r["logits"], "f32", "C2")
    m.shutdown()


@pytest.mark.parametrize("case,B", [(0, 5), (4, 5), (4, 16)])      # B = 16, He = 64: the whole-sequence encoder kernels
def test_bf16_path_runs_close(cuda, case, B):
    """bf16-operand MFMA path (fp32 accumulate, fp32 master copies): logits within tests/tol.py's bf16 bounds (2e-3 max-abs and
    2.5 % of the largest reference logit; measured 5.8e-4)."""
    m, O, ocfg, P, st, batch = make(CASES[case], B=B, W=36, maxlen=6, compute="bf16")
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    print(f"[parity] bf16 logits max-abs {e:.3e}; loss {loss:.4f} vs {float(loss_ref) * B:.4f}")
    check_logits(lg, aux["logits"], "bf16", f"small case {case} B={B}")
    grads = m.get_gradients()
    for k in ("proj.w", "dec.attn.wc", "enc_fw.l1.h2h.w", "enc_bw.l1.h2h.w", "enc_fw.l1.i2h.w", "cnn.conv6.w", "cnn.conv2.b", "cnn.conv4.b",
              "cnn.conv6.b"):
        a, b = grads[k].double().reshape(-1), G[k].double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        print(f"[parity] bf16 grad {k}: rel {relerr(grads[k], G[k]):.3e} cosine {cos:.5f}")
        assert cos > 0.97, k          # tiny-batch BatchNorm + pool arg-max flips make max-rel meaningless in bf16
    m.shutdown()


@pytest.mark.parametrize("B,W", [(6, 72), (3, 200)])
def test_dma_conv_kernel_matches_tiled_kernel(cuda, monkeypatch, B, W):
    """The 256x256x64 LDS-DMA conv kernel (forward + data gradient) against the 128x128 register-staged bf16 kernel on
    the same bf16 operands: only the fp32 accumulation order differs, so features, logits and every gradient must agree
    far inside the bf16-vs-oracle tolerance.  AOCR_FORCE_DMA=1 selects the DMA kernel wherever its shape rules hold
    (it is otherwise chosen only when the grid fills the 256 CUs)."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for force in ("0", "1"):
        monkeypatch.setenv("AOCR_FORCE_DMA", force)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[force] = dict(loss=loss, feats=m.get_tensor("feats").clone(), conv6=m.get_tensor("conv6").clone(),
                          logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["0"], out["1"]
    for k in ("conv6", "feats", "logits"):
        e = (a[k].double() - b[k].double()).abs().max().item()
        print(f"[parity] dma-vs-tiled {k} max-abs {e:.3e}")
        assert e < 2e-3, k        # bf16 re-rounding of activations that differ in the last fp32 bits
    assert abs(a["loss"] - b["loss"]) < 1e-3 * max(1.0, abs(a["loss"]))
    worst = 0.0
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):      # bias in front of a BatchNorm: exact gradient 0, only rounding noise
            continue
        e = relerr(b["grads"][k], a["grads"][k]); worst = max(worst, e)
        assert e < 2e-2, (k, e)
    print(f"[parity] dma-vs-tiled worst gradient rel {worst:.3e}")


@pytest.mark.parametrize("B,W", [(4, 256), (2, 512), (4, 128), (3, 256)])
def test_halo_conv_kernel_matches_im2col_kernel(cuda, monkeypatch, B, W):
    """The halo-resident 3x3 kernel (gemm_halo_bf16_kernel: the input halo of a 32-channel chunk staged once in LDS, taps as
    shifted fragment reads, K chunk-major) against the im2col LDS-DMA kernel (AOCR_NO_HALO=1) on the same bf16 operands, forward
    (plain, (2,1)- and 2x2-pooled row orders) and data gradient: map widths 64 (R = 4 rows per tile), 128 (R = 2) and 32 (R = 8:
    only the 8-row maps qualify).  Only the fp32 accumulation order differs."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    monkeypatch.setenv("AOCR_FORCE_DMA", "1")
    out = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("AOCR_NO_HALO", knob)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[knob] = dict(loss=loss, taps={k: m.get_tensor(k).clone() for k in ("conv3", "conv4", "conv5", "conv6", "feats")},
                         logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    for k in a["taps"]:
        d = (a["taps"][k].double() - b["taps"][k].double()).abs(); e, me = d.max().item(), d.mean().item()
        print(f"[parity] halo-vs-im2col {k} max-abs {e:.3e} mean-abs {me:.3e}")
        assert e < 5e-2 and me < (2e-3 if k == "feats" else 2e-4), (k, e, me)       # the taps are bf16 shadows: a value whose fp32 sums differ in the last bits rounds to the neighbouring bf16 (one ulp of a value < 8), and such flips propagate
    e = (a["logits"].double() - b["logits"].double()).abs().max().item(); print(f"[parity] halo-vs-im2col logits max-abs {e:.3e}"); assert e < 2e-2, e
    assert abs(a["loss"] - b["loss"]) < 1e-3 * max(1.0, abs(a["loss"]))
    assert_grads_agree_up_to_decisions(a["grads"], b["grads"], "halo-vs-im2col")


@pytest.mark.slow
@pytest.mark.parametrize("B,W", [(8, 128), (6, 256), (4, 512)])
def test_halo_four_wave_kernel_matches_eight_wave_kernel(cuda, monkeypatch, B, W):
    """gemm_halo4_bf16_kernel (four waves of 128 x 128, fragment reads / LDS-DMA pieces pinned between the MFMAs, output tiles staged
    through LDS) against the 8-wave gemm_halo_bf16_kernel it replaces (AOCR_HALO8=1) and against its own direct epilogue
    (AOCR_HALO4_STAGED=0): the same k order per output element, so the forward pass is bit-identical; the gradients agree up to
    the order of the fp32 atomics of the split-K weight gradients.  Map widths 32 / 64 / 128 (image widths 128 / 256 / 512): all three
    halo geometries, forward (plain, (2,1)-pooled, fp32 in front of a BatchNorm) and data gradient."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    monkeypatch.setenv("AOCR_FORCE_DMA", "1")                   # small batches: take the 256 x 256 kernels although they do not fill the chip
    out = {}
    # (the staged fp32 tile also leaves the BatchNorm's partial sums: another summation order of the statistics, compared separately below)
    for name, env in (("four", {"AOCR_NO_BN_STATS_FUSE": "1"}), ("eight", {"AOCR_HALO8": "1", "AOCR_NO_BN_STATS_FUSE": "1"}), ("direct", {"AOCR_HALO4_STAGED": "0"}), ("fused", {}), ("y16", {"AOCR_BN_Y16": "1"})):
        for k in ("AOCR_HALO8", "AOCR_HALO4_STAGED", "AOCR_NO_BN_STATS_FUSE", "AOCR_BN_Y16"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=5, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[name] = dict(loss=loss, feats=m.get_tensor("feats").clone(), logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(),
                         taps={k: m.get_tensor(k).clone() for k in ("conv3", "conv4", "conv5", "conv6")}, bn_state=m.bn_state.clone(),
                         dfeats=m.get_tensor("dfeats").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a = out["eight"]
    for name in ("four", "direct"):
        b = out[name]
        assert torch.equal(a["feats"], b["feats"]) and torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"], name
        for k in a["taps"]:
            assert torch.equal(a["taps"][k], b["taps"][k]), (name, k)
        worst = ("", 0.0)
        for k in a["grads"]:
            if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
                continue
            x = relerr(b["grads"][k], a["grads"][k])
            if x > worst[1]: worst = (k, x)
            assert x < 1e-4, (name, k, x)
        print(f"[parity] halo {name}-vs-eight B={B} W={W}: forward bit-identical, worst gradient rel {worst[1]:.3e} ({worst[0]})")
    # BatchNorm statistics from the conv epilogue (per-tile column sums of the stored values, fp32 over 64 rows, fp64 above) against the statistics pass over y
    b = out["fused"]
    e = relerr(b["feats"], a["feats"]); el = (a["logits"].double() - b["logits"].double()).abs().max().item()
    print(f"[parity] BatchNorm statistics in the conv epilogue B={B} W={W}: feats rel {e:.3e}, logits max-abs {el:.3e}")
    assert e < 2e-2 and el < 2e-2 and abs(a["loss"] - b["loss"]) < 1e-3 * max(1.0, abs(a["loss"]))      # bf16 shadows downstream: a last-bit change of a statistic moves values by one bf16 ulp
    es = relerr(b["bn_state"], a["bn_state"])                      # the statistics themselves (running means / variances after this step)
    bn = {k: cosine(b["grads"][k], a["grads"][k]) for k in a["grads"] if k.startswith("cnn.bn")}
    print(f"[parity]   running statistics rel {es:.3e}; BatchNorm weight / bias gradients, cosine:", {k: f"{v:.6f}" for k, v in bn.items()})
    assert es < 1e-5
    # the opt-in bf16 pre-BatchNorm maps (AOCR_BN_Y16=1; off by default: one rounding more than the bf16-operand model): activations and the later layers' statistics within bf16 noise
    c = out["y16"]
    ey = relerr(c["feats"], b["feats"]); esy = relerr(c["bn_state"], b["bn_state"])
    print(f"[parity]   AOCR_BN_Y16=1: feats rel {ey:.3e}, running statistics rel {esy:.3e}, loss {c['loss']:.5f} vs {b['loss']:.5f}")
    assert ey < 3e-2 and esy < 1e-3 and abs(c["loss"] - b["loss"]) < 2e-3 * max(1.0, abs(b["loss"]))
    assert min(bn.values()) > 0.995       # (a handful of images: the single ReLU / arg-max decisions that flip with a one-ulp bf16 change of an activation are visible in these sums)


@pytest.mark.parametrize("He,B,W,Le", [(64, 16, 40, 1), (256, 32, 72, 1), (128, 16, 36, 1), (64, 16, 44, 2), (256, 16, 36, 2),
                                        (64, 16, 800, 1),           # W = 800: T = 199 steps (BASELINE C4 upper width)
                                        (512, 16, 100, 1), (512, 40, 52, 2), (256, 21, 60, 1), (128, 70, 36, 1),   # He = 512; ragged batches
                                        (128, 24, 36, 3)])          # B mod 16 in 1..8 with stacked layers: the 8-row launches' slot offsets (4 ceil(B/16) per layer) exceed 2 ceil(B/8)
def test_seq_encoder_kernels_match_step_kernels(cuda, monkeypatch, He, B, W, Le):
    """The whole-sequence BiLSTM encoder kernels against the per-step kernels on the same bf16 operands (only fp32 summation order
    differs).  Two families: CLUSTER kernels (rnn_cluster.hip: He/64 CUs share 16 rows, recurrent weights resident in registers,
    h(t) / partial d h exchanged as tagged granules; any batch size, He up to 512) and the one-workgroup-per-16-rows kernels
    (rnn_seq.hip, B % 16 == 0, He <= 256; AOCR_NO_CLUSTER=1).  AOCR_NO_SEQ=1 forces the per-step path."""
    cfg = dict(enc_hidden=He, enc_layers=Le, dec_layers=2, input_feed=True)      # Le = 2: the lower layer takes its d h from the layer above
    out = {}
    variants = {"cluster": {}, "step": {"AOCR_NO_SEQ": "1"}}
    if B % 16 == 0 and He <= 256:
        variants["seq"] = {"AOCR_NO_CLUSTER": "1"}
    for name, env in variants.items():          # the new paths first: they must not be able to inherit another run's buffers
        monkeypatch.delenv("AOCR_NO_SEQ", raising=False); monkeypatch.delenv("AOCR_NO_CLUSTER", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        loss2 = m.train_forward_backward(batch)                 # a second step on the same model: the exchange tags move on with the launch epoch
        assert loss2 == pytest.approx(loss, rel=1e-5)
        if name == "cluster":
            assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0, "a cluster kernel timed out waiting for its group"
        out[name] = dict(loss=loss, context=m.get_tensor("context").clone(), logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(),
                         dfeats=m.get_tensor("dfeats").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a = out["step"]
    for name in variants:
        if name == "step":
            continue
        b = out[name]
        for k in ("context", "logits"):
            e = (a[k].double() - b[k].double()).abs().max().item()
            print(f"[parity] {name}-vs-step He={He} B={B} {k} max-abs {e:.3e}")
            assert e < 5e-3, (name, k)  # bf16 re-rounding of h(t) after a different fp32 summation order, compounded over T steps
        assert abs(a["loss"] - b["loss"]) < 2e-3 * max(1.0, abs(a["loss"]))
        e = relerr(b["dfeats"], a["dfeats"]); print(f"[parity] {name}-vs-step dfeats rel {e:.3e}"); assert e < 3e-2
        worst = 0.0
        for k in a["grads"]:
            if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):      # bias in front of a BatchNorm: exact gradient 0, only rounding noise
                continue
            e = relerr(b["grads"][k], a["grads"][k]); worst = max(worst, e)
            assert e < 3e-2, (name, k, e)
        print(f"[parity] {name}-vs-step worst gradient rel {worst:.3e}")


@pytest.mark.parametrize("He,B,W,Le,chunks,p", [(64, 16, 44, 2, 3, 0.0), (256, 16, 100, 2, 4, 0.0), (512, 40, 52, 2, 5, 0.0), (128, 21, 60, 3, 2, 0.0),
                                                 (512, 16, 804, 2, 0, 0.0),          # chunks = 0: the production choice (T = 200: 4 chunks)
                                                 (256, 16, 60, 2, 3, 0.3), (64, 35, 44, 3, 11, 0.2)])   # dropout between the layers; one step per chunk
def test_encoder_layer_wavefront_matches_sequential(cuda, monkeypatch, capfd, He, B, W, Le, chunks, p):
    """Stacked encoder (Le >= 2, BASELINE C5): the layers run as a wavefront of sequence chunks on separate streams (model.hip
    encoder_forward_pipe / encoder_backward_pipe: layer l starts chunk c when layer l-1 has finished it; the cluster kernels take an
    iteration range and pick c / bf16 h / d c / bf16 d z of the iteration before from the state slots) against the same kernels run
    layer after layer (AOCR_NO_LAYER_PIPE=1).  The arithmetic is the same: context and logits bit-identical; gradients up to the order
    of the fp32 atomics of the split-K weight gradients and of the bias sums.  With p > 0 the cluster path also has to agree with the
    per-step kernels (the masked input of layer l is made from the bf16 h of layer l-1: the cluster kernels write no fp32 h)."""
    cfg = dict(enc_hidden=He, enc_layers=Le, dec_layers=2, input_feed=True)
    out = {}
    variants = {"pipe": {"AOCR_LAYER_PIPE_CHUNKS": str(chunks)} if chunks else {}, "seq": {"AOCR_NO_LAYER_PIPE": "1"}}
    if p > 0:
        variants["step"] = {"AOCR_NO_SEQ": "1"}
    for name, env in variants.items():
        for k in ("AOCR_LAYER_PIPE_CHUNKS", "AOCR_NO_LAYER_PIPE", "AOCR_NO_SEQ"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("AOCR_TRACE", "1")
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        if p > 0:
            m.dropout = p; m.global_step = 5
        capfd.readouterr()
        loss = m.train_forward_backward(batch)
        err = capfd.readouterr().err
        assert ("layer wavefront" in err) == (name == "pipe"), err[-2000:]
        loss2 = m.train_forward_backward(batch)                 # again on the same model: new launch epochs, re-recorded events
        if p == 0:
            assert loss2 == pytest.approx(loss, rel=1e-5)       # (p > 0: a repeated step draws another mask, the same one in every variant)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0, "a cluster kernel timed out waiting for its group"
        out[name] = dict(loss=loss2, context=m.get_tensor("context").clone(), logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(),
                         dfeats=m.get_tensor("dfeats").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["seq"], out["pipe"]
    assert torch.equal(a["context"], b["context"]) and torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    e = relerr(b["dfeats"], a["dfeats"]); print(f"[parity] layer wavefront He={He} B={B} Le={Le} chunks={chunks}: dfeats rel {e:.3e}"); assert e < 1e-4
    worst = ("", 0.0)
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        x = relerr(b["grads"][k], a["grads"][k])
        if x > worst[1]: worst = (k, x)
        assert x < 1e-4, (k, x)
    print(f"[parity] layer wavefront vs sequential: worst gradient rel {worst[1]:.3e} ({worst[0]})")
    if p > 0:
        c = out["step"]
        e = (c["context"].double() - a["context"].double()).abs().max().item()
        print(f"[parity] dropout p={p} Le={Le}: cluster vs per-step context max-abs {e:.3e}, loss {a['loss']:.5f} vs {c['loss']:.5f}")
        assert e < 5e-3 and abs(a["loss"] - c["loss"]) < 2e-3 * max(1.0, abs(c["loss"]))
        for k in a["grads"]:
            if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
                continue
            assert relerr(a["grads"][k], c["grads"][k]) < 4e-2, k


@pytest.mark.parametrize("He,B,W", [(256, 8, 416), (256, 8, 800), (256, 4, 1100), (512, 8, 100), (512, 8, 420), (512, 4, 600)])
def test_attention_bf16_general_T_matches_generic(cuda, monkeypatch, He, B, W):
    """attn_bf16_kernel (bf16 context shadow; register-resident slice for T <= 256 at Hd = 512 / T <= 128 at Hd = 1024, two streamed
    passes beyond) against the generic fp32-context kernel (AOCR_NO_ATTN_BF16=1) inside the whole train step: T = 103, 199, 274 at
    Hd = 512; T = 24, 104, 149 at Hd = 1024.  The two differ by the bf16 rounding of the context only."""
    cfg = dict(enc_hidden=He, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_ATTN_BF16", "1")
        else:
            monkeypatch.delenv("AOCR_NO_ATTN_BF16", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dctx=m.get_tensor("dcontext").clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    e = (a["logits"].double() - b["logits"].double()).abs().max().item()
    r = relerr(b["dctx"], a["dctx"])
    print(f"[parity] attention bf16 general-T He={He} W={W}: logits max-abs {e:.3e}, d(context) rel {r:.3e}, loss {b['loss']:.4f} vs {a['loss']:.4f}")
    assert e < 5e-3 and r < 3e-2
    for k in ("dec.attn.wa", "dec.attn.wc", "dec.l2.h2h.w", "enc_fw.l1.h2h.w"):
        assert relerr(b["grads"][k], a["grads"][k]) < 3e-2, k


@pytest.mark.parametrize("B,W,maxlen", [(32, 72, 6), (16, 100, 9), (45, 52, 5), (256, 256, 23), (70, 416, 12), (8, 800, 4)])
def test_decoder_cluster_kernel_matches_launch_chain(cuda, monkeypatch, B, W, maxlen):
    """The teacher-forced decoder loop as ONE launch (dec_cluster.hip: 32 CUs share 32 rows, W1 / W2 / W_c resident in registers,
    out / h1 / h2 / attention context exchanged as tagged granules, scores against the pre-multiplied context ctx . W_a) against
    the per-step launch chain (AOCR_NO_DEC_CLUSTER=1) on the same bf16 operands: Hd = 512, two layers, input feed; ragged
    batches, T up to 199, the C3 shape.  Also the gold-pass decode (the same loop without the saved gates)."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "fwd", "1"):                               # both cluster kernels | forward only (BPTT through the launch chain) | launch chain
        monkeypatch.delenv("AOCR_NO_DEC_CLUSTER", raising=False); monkeypatch.delenv("AOCR_NO_DEC_CLUSTER_BWD", raising=False)
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_DEC_CLUSTER", "1")
        elif knob == "fwd":
            monkeypatch.setenv("AOCR_NO_DEC_CLUSTER_BWD", "1")
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=maxlen, compute="bf16", max_decoder_l=maxlen + 1, max_beam=1)
        loss = m.train_forward_backward(batch)
        loss2 = m.train_forward_backward(batch)
        assert loss2 == pytest.approx(loss, rel=1e-5)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0, "a cluster kernel timed out waiting for its group"
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dctx=m.get_tensor("dcontext").clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        dloss, _ = m.step(batch, True, 1)                       # forward_only: the gold pass runs the same loop without saved gates
        out[knob]["gold"] = [float(x) for x in m._dec_out.gold_scores] + [float(dloss)]
        m.shutdown()
    a = out["1"]
    for name in ("fwd", "0"):
        b = out[name]
        e = (a["logits"].double() - b["logits"].double()).abs().max().item()
        r = relerr(b["dctx"], a["dctx"])
        print(f"[parity] decoder cluster ({name}) B={B} W={W} L={maxlen + 1}: logits max-abs {e:.3e}, d(context) rel {r:.3e}, loss {b['loss']:.5f} vs {a['loss']:.5f}")
        assert e < 1e-2 and r < 3e-2
        assert abs(a["loss"] - b["loss"]) < 2e-3 * max(1.0, abs(a["loss"]))
        worst = ("", 0.0)
        for k in a["grads"]:
            if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
                continue
            x = relerr(b["grads"][k], a["grads"][k])
            if x > worst[1]: worst = (k, x)
            assert x < 3e-2, (name, k, x)
        print(f"[parity] decoder cluster ({name}) worst gradient rel {worst[1]:.3e} ({worst[0]})")
    b = out["0"]
    for x, y in zip(a["gold"], b["gold"]):
        assert abs(x - y) < 2e-2 * max(1.0, abs(x)), (x, y)


@pytest.mark.parametrize("B,W,maxlen", [(32, 72, 6), (16, 100, 9), (45, 52, 5), (7, 72, 3), (256, 256, 23), (70, 416, 12)])
def test_decoder_chain_kernels_match_cluster_kernels_bitwise(cuda, monkeypatch, B, W, maxlen):
    """Round 5 (dec_chain.hip): the decoder loop and its BPTT with TWO interleaved 16-row chains per group and the tag-free exchange
    (pre-filled destinations, no acknowledgement wait, no flags) against dec_cluster.hip's kernels (AOCR_NO_DEC_CHAINS=1).  Same layouts,
    same K split, same reduction order: every output must be BIT-identical -- logits, every gradient, d(context), the gold-pass scores.
    Ragged batches (rows beyond B inside a chain, a whole chain beyond B), several groups, T up to 103, the C3 shape."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("AOCR_NO_DEC_CHAINS", knob)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=maxlen, compute="bf16", max_decoder_l=maxlen + 1, max_beam=1)
        loss = m.train_forward_backward(batch)
        loss2 = m.train_forward_backward(batch)
        assert loss2 == loss
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0, "a whole-sequence kernel timed out waiting for its group"
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), outs=m.get_tensor("outs").clone(),
                         taps={k: m.get_tensor(k).clone() for k in ("ds_all", "dq_all", "dpre_all")},
                         init={k: m.get_tensor(k).clone() for k in ("dh_rec0", "dh_rec1", "dc_st0", "dc_st1", "dfeed0")},
                         dctx=m.get_tensor("dcontext").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        dloss, _ = m.step(batch, True, 1)
        out[knob]["gold"] = [float(x) for x in m._dec_out.gold_scores] + [float(dloss)]
        m.shutdown()
    a, b = out["1"], out["0"]
    assert a["loss"] == b["loss"]
    assert torch.equal(a["logits"], b["logits"]), (a["logits"] - b["logits"]).abs().max().item()
    assert torch.equal(a["outs"], b["outs"])
    for k in a["taps"]:                                          # what the BPTT kernel itself writes, step by step
        assert torch.equal(a["taps"][k], b["taps"][k]), (k, relerr(b["taps"][k], a["taps"][k]))
    # the gradients of the initial decoder state (what the encoder's BPTT starts from): the two kernels are different compilations of the same fp32
    # cell-backward expressions and may contract a multiply-add differently -- an ulp in the running d c, invisible in everything that goes on
    # through bf16 (the taps above), visible here
    for k in a["init"]:
        assert relerr(b["init"][k], a["init"][k]) < 1e-6, k
    assert relerr(b["dctx"], a["dctx"]) < 1e-5
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):      # analytically zero (a bias in front of a BatchNorm): rounding noise only
            continue
        # decoder / projector / embedding: straight from the kernels' outputs.  Encoder and CNN: an ulp in the initial-state gradient is amplified by
        # bf16 operand rounding and ReLU / arg-max decisions on the way down (test_c2_gradient_residual_is_decision_flips): the cluster-vs-chain bound
        tol = 1e-5 if k.startswith(("dec.", "proj", "lookup", "emb")) else 3e-2
        assert relerr(b["grads"][k], a["grads"][k]) < tol, (k, relerr(b["grads"][k], a["grads"][k]))
    assert a["gold"] == b["gold"]


@pytest.mark.parametrize("B,W,maxdec", [(32, 72, 10), (45, 100, 12), (256, 256, 50), (8, 800, 6)])
def test_greedy_decode_cluster_kernel_matches_launch_chain(cuda, monkeypatch, B, W, maxdec):
    """Greedy decode (model.lua:376-536 at beam 1) through the decoder cluster kernel's DEC variant -- cell, attention, projector on fp32
    out, LogSoftMax, selection and the PAD-after-EOS rule inside one launch -- against the launch chain + project_select_kernel.
    Labels must agree except where the two paths' logits tie to within bf16 noise (random weights: near-uniform log-probabilities)."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_DEC_CLUSTER", "1")
        else:
            monkeypatch.delenv("AOCR_NO_DEC_CLUSTER", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=min(5, maxdec - 1), compute="bf16", max_decoder_l=maxdec, max_beam=1)
        # sharpen the output distribution so that the arg-max is not a coin toss between 39 near-equal classes
        P2 = dict(P); P2["proj.w"] = P["proj.w"] * 40.0; m.set_parameters(P2, st)
        loss, stats = m.step(batch, True, 1)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0
        o = m._dec_out
        out[knob] = dict(labels=np.array(o.labels), scores=np.array(o.scores), gold=np.array(o.gold_scores), loss=loss)
        m.shutdown()
    a, b = out["1"], out["0"]
    same_rows = (a["labels"] == b["labels"]).all(axis=1)
    agree = (a["labels"] == b["labels"]).mean()
    print(f"[parity] greedy decode cluster B={B} W={W} Lt={maxdec}: label agreement {agree:.4f}, identical rows {same_rows.mean():.3f}, "
          f"score max-abs on identical rows {np.abs(a['scores'] - b['scores'])[same_rows].max() if same_rows.any() else float('nan'):.3e}")
    assert agree >= 0.97 and same_rows.mean() >= 0.85
    assert np.abs(a["scores"] - b["scores"])[same_rows].max() < 2e-2 * max(1.0, np.abs(a["scores"]).max())
    assert np.abs(a["gold"] - b["gold"]).max() < 2e-2 * max(1.0, np.abs(a["gold"]).max())


@pytest.mark.parametrize("B,W,maxdec,beam", [(32, 72, 10, 5), (45, 100, 12, 3), (256, 256, 50, 5), (16, 800, 30, 5), (7, 100, 40, 2), (40, 160, 16, 8)])
def test_beam_decode_chain_kernel_matches_launch_chain(cuda, monkeypatch, B, W, maxdec, beam):
    """Beam search (model.lua:360-536) inside the decoder chain kernel's BEAM variant -- the k hypotheses of an image are rows of one chain,
    the state gather by parent beam is a lane permutation of the old-state products and cell states, the image's owner runs LogSoftMax and
    project_select_kernel's k-best selection, tokens + parents go to the history beam_backtrace reads -- against the per-step launch chain
    (AOCR_NO_DEC_CHAINS_BEAM=1).  As for greedy decode the two paths' logits differ by bf16 noise, so labels agree except at near-ties."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_DEC_CHAINS_BEAM", "1")
        else:
            monkeypatch.delenv("AOCR_NO_DEC_CHAINS_BEAM", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=min(5, maxdec - 1), compute="bf16", max_decoder_l=maxdec, max_beam=beam)
        P2 = dict(P); P2["proj.w"] = P["proj.w"] * 40.0; m.set_parameters(P2, st)
        loss, stats = m.step(batch, True, beam)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0
        o = m._dec_out
        out[knob] = dict(labels=np.array(o.labels), scores=np.array(o.scores), gold=np.array(o.gold_scores), loss=loss)
        m.shutdown()
    a, b = out["1"], out["0"]
    same_rows = (a["labels"] == b["labels"]).all(axis=1)
    agree = (a["labels"] == b["labels"]).mean()
    print(f"[parity] beam-{beam} decode chain kernel B={B} W={W} Lt={maxdec}: label agreement {agree:.4f}, identical rows {same_rows.mean():.3f}, "
          f"score max-abs on identical rows {np.abs(a['scores'] - b['scores'])[same_rows].max() if same_rows.any() else float('nan'):.3e}")
    assert agree >= 0.95 and same_rows.mean() >= 0.8
    assert np.abs(a["scores"] - b["scores"])[same_rows].max() < 2e-2 * max(1.0, np.abs(a["scores"]).max())
    assert np.abs(a["gold"] - b["gold"]).max() < 2e-2 * max(1.0, np.abs(a["gold"]).max())


@pytest.mark.parametrize("B,W,He", [(48, 100, 256), (21, 72, 256), (24, 160, 512)])
def test_encoder_half_tiles_match_full_tiles(cuda, monkeypatch, B, W, He):
    """Round 5: the encoder cluster kernels give a group 8 batch rows instead of 16 when twice the groups fit the chip (half the output bytes per compute unit;
    columns 8..15 of every MFMA tile repeat column 7 and are neither stored nor published).  Same arithmetic per row: against 16-row groups (AOCR_ENC_RH16=1,
    AOCR_ENC_BWD_RH=16) the encoder output, the logits and the loss must be BIT-identical, the gradients equal up to the order in which the bias sums and the
    split-K partial sums meet; ragged last groups (B = 21) and He = 512 (8 members per group) included."""
    cfg = dict(enc_hidden=He, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("half", "full"):
        monkeypatch.delenv("AOCR_ENC_RH16", raising=False); monkeypatch.delenv("AOCR_ENC_BWD_RH", raising=False)
        if knob == "full":
            monkeypatch.setenv("AOCR_ENC_RH16", "1"); monkeypatch.setenv("AOCR_ENC_BWD_RH", "16")
        else:
            monkeypatch.setenv("AOCR_ENC_BWD_RH", "8")
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=7, compute="bf16", max_decoder_l=8, max_beam=1)
        loss = m.train_forward_backward(batch)
        assert m.cluster_status() == 0
        out[knob] = dict(loss=loss, ctx=m.get_tensor("context").clone(), logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["half"], out["full"]
    assert torch.equal(a["ctx"], b["ctx"]) and torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    worst = 0.0
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        e = relerr(a["grads"][k], b["grads"][k]); worst = max(worst, e)
        assert e < 1e-5, (k, e)
    print(f"[parity] encoder 8-row groups vs 16-row groups B={B} He={He}: context / logits / loss bit-identical, worst gradient difference {worst:.2e}")


@pytest.mark.parametrize("B,W,maxdec,beam", [(8, 100, 48, 5), (12, 72, 40, 3)])
def test_beam_decode_chain_kernel_vs_oracle(cuda, B, W, maxdec, beam):
    """The BEAM variant of the chain kernel against the oracle's own beam search (O.decode_beam: model.lua:360-585 in fp64), not only against the
    launch chain: labels, beam scores, gold scores.  bf16 operands against fp64: rows whose two best candidates are closer than the bf16 noise of
    the logits may differ (sharpened projector: few), everything else must be the reference's decode."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=min(5, maxdec - 1), compute="bf16", max_decoder_l=maxdec, max_beam=beam)
    P2 = dict(P); P2["proj.w"] = P["proj.w"] * 40.0; m.set_parameters(P2, st)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    ref = O.decode_beam(P2, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=maxdec)
    loss, stats = m.step(batch, True, beam)
    assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0
    out = m._dec_out
    rl = ref["labels"].numpy().astype(np.int32)
    same = (np.array(out.labels) == rl).all(axis=1)
    agree = (np.array(out.labels) == rl).mean()
    ds = np.abs(np.array(out.scores) - ref["scores"].numpy())
    print(f"[parity] beam-{beam} chain kernel vs O.decode_beam B={B} Lt={maxdec}: label agreement {agree:.4f}, identical rows {same.mean():.3f}, "
          f"score max-abs on identical rows {ds[same].max() if same.any() else float('nan'):.3e}")
    assert agree >= 0.9 and same.mean() >= 0.7
    assert ds[same].max() < 3e-2 * max(1.0, np.abs(ref["scores"].numpy()).max())
    assert np.abs(np.array(out.gold_scores) - ref["gold_scores"].numpy()).max() < 3e-2 * max(1.0, np.abs(ref["gold_scores"].numpy()).max())
    m.shutdown()


@pytest.mark.parametrize("boost", [60.0, 0.45, 0.3])
def test_greedy_decode_early_exit(cuda, monkeypatch, boost):
    """The greedy cluster kernel leaves its loop once every row of a 32-row group has emitted EOS / PAD (all later steps select PAD at no
    cost, model.lua:448-449) and fills the remaining labels with PAD.  A projector bias that favours EOS makes rows finish early (boost 60:
    every row at step 0; 0.45: the smallest boost that still does; 0.3: no row finishes, no exit); labels, scores and gold scores must equal the run that takes all max_decoder_l steps."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_DEC_EARLY", "1")
        else:
            monkeypatch.delenv("AOCR_NO_DEC_EARLY", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=70, W=72, maxlen=5, compute="bf16", max_decoder_l=20, max_beam=1)
        P2 = dict(P); P2["proj.w"] = P["proj.w"] * 40.0
        b2 = P["proj.b"].clone(); b2[2] += boost; P2["proj.b"] = b2          # class index 2 = token 3 = EOS
        m.set_parameters(P2, st)
        loss, stats = m.step(batch, True, 1)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0
        o = m._dec_out
        out[knob] = dict(labels=np.array(o.labels), scores=np.array(o.scores), gold=np.array(o.gold_scores), loss=loss)
        m.shutdown()
    a, b = out["1"], out["0"]
    lens = (b["labels"] != 1).sum(axis=1)
    print(f"[parity] greedy early exit boost={boost}: emitted lengths min {lens.min()} max {lens.max()} of 20 steps")
    assert (a["labels"] == b["labels"]).all()
    assert np.array_equal(a["scores"], b["scores"]) and np.array_equal(a["gold"], b["gold"]) and a["loss"] == b["loss"]
    if boost > 10: assert lens.max() < 20                             # every row finished at once: the exit path was certainly taken


@pytest.mark.parametrize("case,p", [(dict(enc_hidden=32, enc_layers=2, dec_layers=2, input_feed=True), 0.3),
                                    (dict(enc_hidden=32, enc_layers=1, dec_layers=3, input_feed=False), 0.5),
                                    (dict(enc_hidden=64, enc_layers=2, dec_layers=2, input_feed=True), 0.1)])
def test_dropout_train_step_vs_oracle(cuda, case, p):
    """nn.Dropout(p) of LSTM.lua:68-69 (input of every LSTM layer above the first) and :116-118 (attention output) in the training step:
    counter-based masks (seed, train step, site, element), the same function in the kernels' epilogues and in the oracle
    (`dropout_state`); loss and all gradients against autograd through the fp64 oracle (fp32 tolerances).  The decode / forward_only
    step of the same model is unaffected (evaluate())."""
    m, O, ocfg, P, st, batch = make(case, B=5, W=44, maxlen=5, max_decoder_l=8)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    ref_eval = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=1, max_decoder_l=8)
    m.dropout = p; m.global_step = 3
    with O.dropout_state(p, m.dropout_seed, 3):
        loss_ref, G, r, _ = O.train_step_autograd(P, st, ocfg, img.double(), tgt, tge)
    loss = m.train_forward_backward(batch)
    B = img.shape[0]
    print(f"[parity] dropout p={p}: loss {loss:.6f} vs {float(loss_ref) * B:.6f}")
    assert abs(loss - float(loss_ref) * B) < 1e-4 * max(1.0, abs(float(loss_ref) * B))
    got = m.get_gradients(); worst = ("", 0.0)
    for k, g in G.items():
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        e = relerr(got[k], g)
        if e > worst[1]: worst = (k, e)
        # the LSTM / attention / projector tensors (everything dropout touches) to fp32 accuracy; the early CNN may carry a single
        # ReLU / max-pool decision flip between the fp32 kernels and the fp64 oracle, unrelated to dropout
        assert e < (5e-2 if k.startswith("cnn.") else 2e-4), (k, e)
    print(f"[parity] dropout p={p}: worst gradient rel {worst[1]:.3e} ({worst[0]})")
    with O.dropout_state(p, m.dropout_seed, 4):                       # another step number: another mask
        loss_ref2, _, _, _ = O.train_step_autograd(P, st, ocfg, img.double(), tgt, tge)
    m.global_step = 4
    loss2 = m.train_forward_backward(batch)
    assert abs(loss2 - float(loss_ref2) * B) < 1e-4 * max(1.0, abs(float(loss_ref2) * B)) and abs(loss2 - loss) > 1e-6
    m.set_parameters(P, st)                                             # (the training passes moved the running statistics)
    _, stats = m.step(batch, True, 1)                                   # evaluate(): no dropout
    assert np.array_equal(m._dec_out.labels, ref_eval["labels"].numpy().astype(np.int32))
    m.shutdown()


@pytest.mark.parametrize("p,B,W,maxlen", [(0.25, 45, 52, 5), (0.5, 32, 72, 6), (0.1, 70, 100, 9)])
def test_dropout_cluster_kernels_match_launch_chain(cuda, monkeypatch, p, B, W, maxlen):
    """nn.Dropout(p > 0) inside the decoder cluster kernels (round 3: the masks of LSTM.lua:68-69,116-118 are evaluated in the gate /
    attention epilogues of dec_cl_fwd_kernel<false, true> and dec_cl_bwd_kernel<true>) against the per-step launch chain
    (AOCR_NO_DEC_CLUSTER_DROP=1) under the SAME counter-based masks: loss, logits, d(context) and every gradient tensor."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("cluster", "chain"):
        monkeypatch.delenv("AOCR_NO_DEC_CLUSTER_DROP", raising=False)
        if knob == "chain":
            monkeypatch.setenv("AOCR_NO_DEC_CLUSTER_DROP", "1")
        monkeypatch.setenv("AOCR_TRACE", "1")
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=maxlen, compute="bf16", max_decoder_l=maxlen + 1, max_beam=1)
        m.dropout = p; m.global_step = 11
        loss = m.train_forward_backward(batch)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dctx=m.get_tensor("dcontext").clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.global_step = 12
        out[knob]["loss2"] = m.train_forward_backward(batch)      # another step: another mask
        m.shutdown()
    a, b = out["chain"], out["cluster"]
    e = (a["logits"].double() - b["logits"].double()).abs().max().item()
    r = relerr(b["dctx"], a["dctx"])
    print(f"[parity] dropout p={p} cluster vs chain B={B} W={W}: logits max-abs {e:.3e}, d(context) rel {r:.3e}, loss {b['loss']:.5f} vs {a['loss']:.5f}")
    assert e < 2e-2 and r < 3e-2
    assert abs(a["loss"] - b["loss"]) < 2e-3 * max(1.0, abs(a["loss"])) and abs(a["loss2"] - b["loss2"]) < 2e-3 * max(1.0, abs(a["loss2"]))
    assert abs(a["loss"] - a["loss2"]) > 1e-4                     # the mask depends on the step
    worst = ("", 0.0)
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        x = relerr(b["grads"][k], a["grads"][k])
        if x > worst[1]: worst = (k, x)
        assert x < 4e-2, (k, x)
    print(f"[parity] dropout cluster vs chain: worst gradient rel {worst[1]:.3e} ({worst[0]})")


def test_dropout_bf16_cluster_shape(cuda):
    """Dropout at the shape the decoder cluster kernels serve (Hd = 512, bf16; since round 3 THROUGH those kernels): loss against the
    fp64 oracle within the bf16 tolerance, the decoder's gradients by cosine."""
    case = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    m, O, ocfg, P, st, batch = make(case, B=6, W=52, maxlen=5, compute="bf16", max_decoder_l=8, max_beam=1)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    m.dropout = 0.25; m.global_step = 7
    with O.dropout_state(0.25, m.dropout_seed, 7):
        loss_ref, G, r, _ = O.train_step_autograd(P, st, ocfg, img.double(), tgt, tge)
    loss = m.train_forward_backward(batch)
    B = img.shape[0]
    print(f"[parity] dropout bf16: loss {loss:.5f} vs {float(loss_ref) * B:.5f}")
    assert abs(loss - float(loss_ref) * B) < 2e-3 * abs(float(loss_ref) * B)
    got = m.get_gradients()
    for k in ("dec.l1.i2h.w", "dec.l2.i2h.w", "dec.l2.h2h.w", "dec.attn.wc", "proj.w", "enc_fw.l1.h2h.w"):
        a, b = got[k].double().flatten(), G[k].double().flatten()
        cos = float((a @ b) / (a.norm() * b.norm()))
        assert cos > 0.999, (k, cos)
    m.shutdown()


def test_c3_full_size_kernel_paths_agree(cuda, monkeypatch):
    """BASELINE config C3 at full size (32x256, B=256, He=256, L=24, bf16): the production dispatch (256x256 LDS-DMA conv /
    filter-gradient kernels, whole-sequence encoder kernels -- chosen by shape, no forcing) against the 128x128 and
    per-step kernels (AOCR_NO_DMA=1, AOCR_NO_SEQ=1) on the same seeded batch and weights.  Size-independent property: the
    two kernel families compute the same contractions, so logits / loss / every gradient agree to fp32-summation-order
    noise amplified by bf16 re-rounding."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("AOCR_NO_DMA", knob); monkeypatch.setenv("AOCR_NO_SEQ", knob)
        m, O, ocfg, P, st, batch = make(cfg, B=256, W=256, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[knob] = dict(loss=loss, feats=m.get_tensor("feats").clone(), logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    # (the halo-resident conv kernel sums K chunk-major, the tiled kernels tap-major: bf16 shadows flip by an ulp where the fp32 sums differ in the last bits)
    d = (a["feats"].double() - b["feats"].double()).abs(); e, me = d.max().item(), d.mean().item()
    print(f"[parity] C3 full size: feats max-abs {e:.3e} mean-abs {me:.3e}"); assert e < 5e-2 and me < 2e-3
    e = (a["logits"].double() - b["logits"].double()).abs().max().item(); print(f"[parity] C3 full size: logits max-abs {e:.3e}"); assert e < 5e-3
    assert abs(a["loss"] - b["loss"]) < 2e-3 * abs(a["loss"])
    assert_grads_agree_up_to_decisions(a["grads"], b["grads"], "C3 full size")


def test_logits_c3_shape_bf16(cuda):
    """BASELINE config C3 at full size through the production bf16 dispatch: decoder logits against the fp64 oracle
    (tests/tol.py: 2e-3 max-abs and 2.5 % of the largest reference logit; measured 4.2e-4; fp32 accumulation, fp32 master copies)."""
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=256, W=256, maxlen=23,
                                    compute="bf16", max_decoder_l=24, max_beam=1)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    with torch.no_grad():
        r = O.forward_train(P, {k: v.clone() for k, v in st.items()}, ocfg, img, tgt, tge, training=True)
    logits, loss = m.forward_logits(batch, training=True)
    e = (logits.double() - r["logits"]).abs().max().item()
    print(f"[parity] C3 bf16 logits max-abs {e:.3e} (mean |logit| {r['logits'].abs().mean().item():.3f}, max {r['logits'].abs().max().item():.3f}); loss {loss:.4f} vs {float(r['loss']) * 256:.4f}")
    check_logits(logits, r["logits"], "bf16", "C3 full size")
    assert abs(loss - float(r["loss"]) * 256) < 2e-3 * abs(loss)
    m.shutdown()


def test_train_step_parity_c2_shape(cuda):
    """BASELINE config C2 at full size (32x100, B=64, He=256, Ld=2, input feed, L=24), exact-fp32 MFMA mode: loss and EVERY
    gradient tensor of the fused train step against the fp64 oracle's hand-ordered BPTT (the small-shape test checks the
    same at B=5)."""
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=64, W=100, maxlen=23,
                                    max_decoder_l=24, max_beam=1)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    loss_ref, G, aux, st_new = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    assert abs(loss - float(loss_ref) * 64) < 1e-4 * abs(loss)
    grads = m.get_gradients()
    worst = ("", 0.0); bad = []
    for k, g in G.items():
        e = relerr(grads[k], g)
        if g.abs().max() < 1e-9:                         # conv biases in front of BatchNorm have ~0 gradient
            e = (grads[k].double() - g).abs().max().item()
        a, b = grads[k].double().reshape(-1), g.double().reshape(-1)
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-300)) if b.norm() > 1e-9 else 1.0
        if e > worst[1]: worst = (k, e)
        print(f"[parity] C2 grad {k:22s} rel {e:.3e} cosine {cos:.7f}")
        # Everything from conv7 upwards must match to fp32 accuracy.  Below the (2,1) pool after conv6 a handful of the 1.6 M
        # pooling windows are fp32-vs-fp64 near-ties (arg-max) or sit within rounding of the ReLU threshold; each flip moves ONE
        # term of a random-sign sum over ~51 k pixels, which is a percent-level change of single weight-gradient entries
        # (max-norm) but leaves the direction untouched -- so those layers are held to the cosine.
        early = k.startswith("cnn.") and not k.startswith(("cnn.conv7", "cnn.bn7"))
        if (early and (cos < 0.9995 or e > 5e-2)) or (not early and e > 2e-3):
            bad.append((k, e, cos))
    assert not bad, bad
    print(f"[parity] C2 full size fp32: loss {loss:.4f} vs {float(loss_ref) * 64:.4f}; {len(G)} gradient tensors, worst rel {worst[1]:.3e} ({worst[0]})")
    m.shutdown()


def _gpu_cnn_decisions(m, B):
    """The ReLU / max-pool decisions of the GPU's last forward pass as the oracle's `cnn_decisions` (conv2..conv7; conv1's pooling
    stores no arg-max -- its backward recomputes the window -- and stays with the oracle's own choice)."""
    dec = {}
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous()
    for i in (2, 4, 6):
        dec[f"idx{i}"] = nchw(m.get_tensor(f"idx{i}")).long()
        dec[f"act{i}"] = nchw(m.get_tensor(f"conv{i}")) > 0
    for i in (3, 5):
        dec[f"act{i}"] = nchw(m.get_tensor(f"conv{i}")) > 0
    feats = m.get_tensor("feats")                                   # (T, B, 512) time-major
    dec["act7"] = (feats > 0).permute(1, 2, 0).unsqueeze(2).contiguous()          # (B, 512, 1, T)
    return dec


def test_c2_gradient_residual_is_decision_flips(cuda):
    """C2 at full size, exact-fp32 mode.  test_train_step_parity_c2_shape holds the early CNN gradients only to a cosine because a
    handful of the 1.6 M pooling windows / ReLU inputs are fp32-vs-fp64 near-ties.  Here that explanation is TESTED: the GPU's own
    decisions (stored arg-max maps, signs of its outputs) are imposed on the fp64 oracle, and then EVERY gradient tensor -- the early
    convolutions included -- agrees to fp32 accuracy; the number of decisions that differed from the oracle's own is reported."""
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=64, W=100, maxlen=23,
                                    max_decoder_l=24, max_beam=1)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    loss = m.train_forward_backward(batch)
    grads = m.get_gradients()
    dec = _gpu_cnn_decisions(m, 64)
    # how many decisions differ from the oracle's own forward pass
    import torch.nn.functional as F
    x = (img.double() - 128.0) / 128.0; st2 = {k: v.clone() for k, v in st.items()}; flips = {}; last = 0; pre = None
    for l in O.CNN_LAYERS:
        if l[0] == "conv":
            _, i, cin, cout, k, pad = l; x = F.conv2d(x, P[f"cnn.conv{i}.w"], P[f"cnn.conv{i}.b"], padding=pad); last = i; pre = x
        elif l[0] == "relu":
            x = F.relu(x)
        elif l[0] == "pool":
            kh, kw = l[1], l[2]; Bq, Cq, Hq, Wq = pre.shape
            win = pre.reshape(Bq, Cq, Hq // kh, kh, Wq // kw, kw).permute(0, 1, 2, 4, 3, 5).reshape(Bq, Cq, Hq // kh, Wq // kw, kh * kw)
            if f"idx{last}" in dec:
                act = win.max(4).values > 0
                flips[f"pool{last}"] = int(((win.argmax(4) != dec[f"idx{last}"]) & act).sum()); flips[f"relu{last}"] = int((act != dec[f"act{last}"]).sum())
            x = F.max_pool2d(x, (kh, kw), (kh, kw))
        elif l[0] == "bn":
            i = l[1]
            x = F.batch_norm(x, st2[f"cnn.bn{i}.rm"], st2[f"cnn.bn{i}.rv"], P[f"cnn.bn{i}.w"], P[f"cnn.bn{i}.b"], training=True, momentum=0.1, eps=1e-5)
            flips[f"relu{i}"] = int(((x > 0) != dec[f"act{i}"]).sum())
    print(f"[parity] C2 decisions that differ between the fp32 kernels and the fp64 oracle: {flips}")
    loss_ref, G, r, _ = O.train_step_autograd(P, st, ocfg, img.double(), tgt, tge, cnn_decisions=dec)
    assert abs(loss - float(loss_ref) * 64) < 1e-4 * abs(loss)
    worst = ("", 0.0)
    for k, g in G.items():
        if g.abs().max() < 1e-9:
            continue
        e = relerr(grads[k], g)
        if e > worst[1]: worst = (k, e)
        print(f"[parity] C2 grad, GPU decisions imposed on the oracle: {k:22s} rel {e:.3e}")
        assert e < 5e-5, (k, e)                                         # (measured <= 7e-6, conv1 included: its own pool decisions did not differ)
    print(f"[parity] C2 with imposed decisions: worst rel {worst[1]:.3e} ({worst[0]})")
    m.shutdown()


@pytest.mark.parametrize("beam", [1, 3])
def test_decode_parity_c2_shape(cuda, beam):
    """BASELINE config C2 at full size, eval-mode BatchNorm, 50 decode steps + gold pass: labels, beam scores, gold scores and
    the gold-pass loss against the fp64 oracle's decode (model.lua:321-627)."""
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=64, W=100, maxlen=23,
                                    max_decoder_l=50, max_beam=3)
    st = {k: (v + 0.05 if k.endswith("rm") else v * 1.3) for k, v in st.items()}
    m.set_parameters(P, st)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    ref = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=50)
    loss, stats = m.step(batch, True, beam)
    out = m._dec_out
    same = (out.labels == ref["labels"].numpy().astype(np.int32)).all(axis=1)
    print(f"[parity] C2 decode beam {beam}: {int(same.sum())}/64 label sequences identical; loss {loss:.4f} vs {float(ref['loss']):.4f}; "
          f"scores max-abs {np.abs(out.scores - ref['scores'].numpy())[same].max():.2e}")
    assert same.sum() >= 63                      # a near-tie between two tokens may flip one sequence in fp32
    assert np.abs(out.scores - ref["scores"].numpy())[same].max() < 5e-3
    assert np.abs(out.gold_scores - ref["gold_scores"].numpy()).max() < 5e-3
    assert abs(loss - float(ref["loss"])) < 2e-3 * max(1.0, float(ref["loss"]))
    m.shutdown()


@pytest.mark.parametrize("wd", [0.0, 1e-3])
def test_adadelta_steps_match_oracle(cuda, wd):
    """optim.adadelta_list (optim_adadelta.lua:19-62) over three consecutive steps, state carried: the fused HIP update against the
    restated tensor ops in fp64, both fed the gradients the HIP step produced (this isolates the update rule)."""
    m, O, ocfg, P, st, batch = make(CASES[0], B=4, W=36, maxlen=5)
    ref = {k: v.double().clone() for k, v in P.items()}
    state = {}
    for step in range(3):
        m.train_forward_backward(batch)
        G = {k: v.double() for k, v in m.get_gradients().items()}
        before = {k: v.double() for k, v in m.get_parameters().items()}
        m.adadelta_step(rho=0.9, eps=1e-6, weight_decay=wd)
        ref = O.adadelta_list(before, G, state, 0.9, 1e-6, wd)
        got = m.get_parameters()
        worst = max((got[k].double() - ref[k]).abs().max().item() for k in ref)
        moved = max((got[k].double() - before[k]).abs().max().item() for k in ref)
        print(f"[parity] adadelta wd={wd} step {step}: max-abs {worst:.3e} (largest update {moved:.3e})")
        assert worst < 2e-6 and moved > 1e-4
    n = m.num_params
    var = m.adadelta_state[:n].cpu(); acc = m.adadelta_state[n:].cpu()
    for name, group, off, shape in m.table:
        k = int(np.prod(shape))
        sv = state[name]["var"]; sa = state[name]["acc"]
        if len(shape) == 4:
            sv = sv.permute(0, 2, 3, 1); sa = sa.permute(0, 2, 3, 1)
        assert (var[off:off + k].double() - sv.reshape(-1)).abs().max().item() < 1e-6 * max(1.0, sv.abs().max().item()), name
        assert (acc[off:off + k].double() - sa.reshape(-1)).abs().max().item() < 1e-7, name
    m.shutdown()


def test_overlapped_exchange_waits_for_gradient_buckets(cuda, monkeypatch):
    """The bucketed data-parallel exchange on ONE GPU with a stand-in collective (x2 in place): every bucket must be complete when
    the communication stream touches it, i.e. the result is exactly twice the gradients of the same step without the exchange."""
    import aocr
    from aocr import check, lib, ptr
    from aocr import dist as adist
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=32, W=100, maxlen=10,
                                    max_decoder_l=12, max_beam=1, compute="bf16")
    images, targets, targets_eval = m._upload(batch)
    B, _, _, W = images.shape
    loss = torch.zeros(1, device=cuda)

    def enqueue():
        check(lib.aocr_train_forward_backward(m._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, targets.shape[1], 1.0 / B, ptr(loss)))
    check(lib.aocr_model_set_stream(m._h, m._stream()))
    enqueue(); torch.cuda.synchronize()
    g0 = m.grad_params.clone(); l0 = loss.clone()
    ranges = adist.bucket_ranges(m.ccfg)
    assert sorted(ranges)[0][0] == 0 and sorted(ranges)[-1][1] == m.num_params
    monkeypatch.setattr(adist, "world_size", lambda: 2)
    monkeypatch.setattr(torch.distributed, "all_reduce", lambda t: t.mul_(2.0))
    comm = torch.cuda.Stream()
    for _ in range(3):
        enqueue()
        adist.exchange_overlapped(m.grad_params, loss, ranges, m._wait_bucket, comm)
        g = m.grad_params.clone()                     # on the main stream, which has joined the communication stream
        torch.cuda.synchronize()
        # split-K atomics make two runs of the same step differ in the last bits: compare with a tolerance far below a missed bucket
        assert (g - 2 * g0).abs().max().item() <= 1e-4 * g0.abs().max().item()
        assert abs(loss.item() - 2 * l0.item()) <= 1e-5 * abs(l0.item())
    m.shutdown()


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_tall_strips_two_layer_encoder_beam5(cuda, compute):
    """The structure of BASELINE.json configs[4] at test size: line strips taller than 32 (the CNN leaves Ho7 > 1 rows, View(512,-1)
    of cnn.lua:44 strings them row-major into the sequence), a 2-layer BiLSTM encoder and beam-width-5 decoding."""
    import aocr
    import oracle_torch as O
    B, H, W = 4, 64, 40                                  # -> 3 x 9 feature positions, T = 27
    ocfg = O.OcrConfig(enc_hidden=64, enc_layers=2, dec_layers=2, input_feed=True)
    P, st = O.init_params(ocfg, 11), O.init_bn_state()
    st = {k: (v + 0.05 if k.endswith("rm") else v * 1.3) for k, v in st.items()}
    img, tgt, tge, nnz = O.synth_batch(B, W, max_len=5, min_len=2, H=H)
    m = aocr.Model()
    m._set_structure(dict(encoder_num_hidden=64, encoder_num_layers=2, decoder_num_layers=2, input_feed=True))
    m._set_runtime(dict(batch_size=B, img_h=H, max_img_w=W, max_decoder_l=10, max_beam=5, compute=compute))
    m.optim_state = {"learningRate": 0.1}
    m._build()
    m.set_parameters(P, st)
    batch = [img, tgt, tge, nnz, [f"img{i}" for i in range(B)]]
    ti, tt, te = (torch.from_numpy(np.asarray(x)) for x in (img, tgt, tge))
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, ti, tt, te)
    loss = m.train_forward_backward(batch)
    assert aux["context"].shape[1] == 27
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    print(f"[parity] tall strips {compute}: T={aux['context'].shape[1]} logits max-abs {e:.3e}; loss {loss:.4f} vs {float(loss_ref) * B:.4f}")
    check_logits(lg, aux["logits"], compute, "tall strips")
    if compute == "f32":
        grads = m.get_gradients()
        for k in ("cnn.conv7.w", "cnn.conv4.w", "enc_fw.l2.i2h.w", "enc_bw.l1.h2h.w", "dec.attn.wa"):
            r = relerr(grads[k], G[k]); assert r < 2e-3, (k, r)
        m.set_parameters(P, st)
        ref = O.decode_beam(P, st, ocfg, ti, tt, te, beam=5, max_decoder_l=10)
        _, stats = m.step(batch, True, 5)
        assert np.array_equal(m._dec_out.labels, ref["labels"].numpy().astype(np.int32))
        assert np.abs(m._dec_out.scores - ref["scores"].numpy()).max() < 2e-3
    m.shutdown()


@pytest.mark.parametrize("compute,B,W", [("bf16", 256, 256), ("f32", 64, 100)])
def test_full_size_properties(cuda, compute, B, W):
    """Size-independent properties at BASELINE's full sizes (C3 bf16, C2 fp32), no oracle needed:
    (1) the backward pass is LINEAR in d(loss): scaling the loss gradient by 4 scales every parameter gradient by 4 (a power of two:
        exact in floating point up to the order of the split-K atomics);
    (2) the step is EQUIVARIANT under a permutation of the batch: logits permute with the images, the loss and the parameter
        gradients do not change (BatchNorm statistics, the loss and every weight gradient are sums over the batch)."""
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=B, W=W, maxlen=23,
                                    compute=compute, max_decoder_l=24, max_beam=1)
    img, tgt, tge, nnz, names = batch
    loss1 = m.train_forward_backward(batch, grad_scale=1.0 / B)
    g1 = {k: v.clone() for k, v in m.get_gradients().items()}
    lg1 = m.get_tensor("logits")[:, :, :ocfg.vocab].clone()
    loss4 = m.train_forward_backward(batch, grad_scale=4.0 / B)
    g4 = m.get_gradients()
    assert loss4 == pytest.approx(loss1, rel=1e-6)
    noisy = ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b")          # bias in front of BatchNorm: the gradient is rounding noise only
    worst = max(relerr(g4[k], 4 * g1[k]) for k in g1 if k not in noisy)
    print(f"[property] {compute} B={B}: linearity in d(loss): worst rel {worst:.2e}")
    assert worst < (2e-3 if compute == "bf16" else 2e-4)
    perm = np.random.default_rng(0).permutation(B)
    pbatch = [np.asarray(img)[perm], np.asarray(tgt)[perm], np.asarray(tge)[perm], nnz, [names[i] for i in perm]]
    lossp = m.train_forward_backward(pbatch, grad_scale=1.0 / B)
    gp = m.get_gradients()
    lgp = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lgp - lg1[:, perm]).abs().max().item()
    worst = max(relerr(gp[k], g1[k]) for k in g1 if k not in noisy)
    print(f"[property] {compute} B={B}: batch permutation: logits max-abs {e:.2e}, loss {lossp:.3f} vs {loss1:.3f}, worst gradient rel {worst:.2e}")
    assert e < (2e-2 if compute == "bf16" else 2e-5)
    assert lossp == pytest.approx(loss1, rel=(1e-4 if compute == "bf16" else 1e-6))
    assert worst < (3e-2 if compute == "bf16" else 1e-3)
    m.shutdown()


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_cluster_kernels_remote_exchange_mode(cuda, monkeypatch, p):
    """The whole-sequence kernels assume that the workgroups of a group land on ONE XCD (round-robin dispatch), check it at run time
    through HW_REG_XCC_ID and otherwise exchange with write-through (sc0 sc1) stores and system-scope loads -- the path a partitioned
    device or an unexpected placement takes.  AOCR_CL_REMOTE=1 forces that path: the same arithmetic, so loss and logits must be
    IDENTICAL to the in-XCD mode (the gradients up to the order of the split-K sums), encoder and decoder, training (with and without
    dropout) and greedy decode."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for mode in ("local", "remote"):
        monkeypatch.delenv("AOCR_CL_REMOTE", raising=False)
        if mode == "remote":
            monkeypatch.setenv("AOCR_CL_REMOTE", "1")
        m, O, ocfg, P, st, batch = make(cfg, B=70, W=100, maxlen=9, compute="bf16", max_decoder_l=12, max_beam=1)
        m.dropout = p; m.global_step = 5
        loss = m.train_forward_backward(batch)
        assert m.cluster_status() == 0
        logits = m.get_tensor("logits")[:, :, :ocfg.vocab].clone()
        grads = {k: v.clone() for k, v in m.get_gradients().items()}
        m.dropout = 0.0
        dl, _ = m.step(batch, True, 1)
        out[mode] = (loss, logits, grads, m._dec_out.labels.copy(), m._dec_out.scores.copy(), dl)
        m.shutdown()
    a, b = out["local"], out["remote"]
    assert a[0] == b[0] and torch.equal(a[1], b[1]), (a[0], b[0], (a[1] - b[1]).abs().max().item())
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and a[5] == b[5]
    worst = max(relerr(b[2][k], a[2][k]) for k in a[2] if k not in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"))
    print(f"[parity] cluster kernels, remote exchange mode (dropout {p}): loss and logits identical, worst gradient rel {worst:.2e}")
    assert worst < 1e-4


def test_update_is_skipped_when_a_cluster_kernel_timed_out(cuda):
    """include/aocr.h (aocr_cluster_status): a whole-sequence kernel whose bounded wait expired leaves a non-zero code; the step's
    gradients are invalid, and aocr_sgd_step / aocr_adadelta_step must then leave the parameters untouched (device-side predicate, no
    host sync) for THAT step.  The code is injected here (a real timeout needs a co-tenant)."""
    import ctypes as C
    import aocr
    from aocr import check, lib
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=32, W=72, maxlen=5, compute="bf16",
                                    max_decoder_l=8, max_beam=1)
    m.train_forward_backward(batch)
    p = C.c_void_p(); nd = C.c_int32(); shape = (C.c_int64 * 4)()
    check(lib.aocr_get_tensor(m._h, b"cl_err", C.byref(p), C.byref(nd), shape))
    off = p.value - m.workspace.data_ptr()
    flag = m.workspace[off:off + 4].view(torch.int32)
    before = m.params.clone()
    bn0 = m.bn_state.clone()
    m.train_forward_backward(batch)                                 # a step whose kernels "time out": its CNN forward moves the running statistics
    bn_moved = m.bn_state.clone()
    assert not torch.equal(bn_moved, bn0)
    flag.fill_(23)                                                  # "dq raise/wait timed out"
    m.sgd_step(lr=0.1)
    torch.cuda.synchronize()
    assert torch.equal(m.params, before), "the update ran on gradients flagged invalid"
    # the optimizer call that skips the update also takes the step's move of the running statistics back (device side),
    # so a repeat of the batch moves them exactly once whenever the host polls the status -- and a host that never repeats loses nothing else
    assert torch.equal(m.bn_state, bn0), "the skipped step's move of the BatchNorm running statistics was not taken back"
    # round 6: the optimizer call CONSUMES the code -- only the step that timed out is skipped, however late the host polls
    # (the code waits for the host in a sticky word); before, every step up to the poll was dropped
    assert int(flag.item()) == 0, "the skipping optimizer call did not consume the step's code"
    m.train_forward_backward(batch)                                 # the host has not polled yet: a healthy step
    torch.cuda.synchronize()
    assert torch.equal(m.bn_state, bn_moved)
    m.sgd_step(lr=0.1)
    torch.cuda.synchronize()
    after1 = m.params.clone()
    assert not torch.equal(after1, before), "a healthy step behind an unpolled time-out was dropped"
    assert m.cluster_status() == 23 and m.cluster_status() == 0     # the late poll still reports the skipped step; read and clear
    # the same through the fused Adadelta pass
    m.train_forward_backward(batch); bn1 = m.bn_state.clone()
    m.train_forward_backward(batch)
    flag.fill_(24)
    m.adadelta_step()
    torch.cuda.synchronize()
    assert torch.equal(m.params, after1) and torch.equal(m.bn_state, bn1)
    m.train_forward_backward(batch); m.adadelta_step()
    torch.cuda.synchronize()
    assert not torch.equal(m.params, after1)
    assert m.cluster_status() == 24 and m.cluster_status() == 0
    # a code no optimizer call consumed (a decode call's time-out: decode never updates) must not cancel the NEXT training step: the step's
    # prologue moves it to the sticky word
    flag.fill_(31)
    p0 = m.params.clone()
    m.train_forward_backward(batch); m.sgd_step(lr=0.1)
    torch.cuda.synchronize()
    assert not torch.equal(m.params, p0), "a stale decode-call code cancelled a healthy training step"
    assert m.cluster_status() == 31 and m.cluster_status() == 0
    with pytest.raises(RuntimeError):
        flag.fill_(11); m.check_health()
    m.shutdown()


@pytest.mark.parametrize("B,W", [(6, 72), (3, 200), (32, 256)])
def test_dma128_kernel_matches_tiled_kernel(cuda, monkeypatch, B, W):
    """Round 4: gemm_dma128_kernel (128 x 128 tiles on the LDS-DMA ring: conv forward / data gradient and the hoisted bf16 GEMMs at the
    shapes the 256-row kernels do not take -- every small-batch test shape, 32-64 lines per GPU) against the register-staged 128 x 128
    kernel it replaces there (AOCR_NO_DMA128=1).  Same operands, same k order per output element: features, logits, loss and every
    gradient must be BIT-identical.  (32, 256) is the strong-scaling slice of C3: both ring depths (one / two workgroups per CU) run."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for off in ("1", "0"):
        monkeypatch.setenv("AOCR_NO_DMA128", off)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, feats=m.get_tensor("feats").clone(), conv6=m.get_tensor("conv6").clone(),
                        logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    assert torch.equal(a["conv6"], b["conv6"]) and torch.equal(a["feats"], b["feats"]) and torch.equal(a["logits"], b["logits"])
    assert a["loss"] == b["loss"]
    for k in a["grads"]:
        if k.endswith(".w") and k.startswith("cnn.conv"):
            # filter gradients: split-K partial sums meet in a different order from run to run (atomics on the small shapes): not bit-stable even between two runs of ONE path
            assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k
        elif k not in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):      # (a bias in front of a BatchNorm: exact gradient 0, rounding noise only)
            assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k            # (max-norm relative: an element-wise bound with atol 1e-7 can trip on an element that is ~0 by cancellation, whose split-K atomics arrive in another order)
    print(f"[parity] dma128 vs register-staged 128 x 128 kernel, B={B} W={W}: conv6 / feats / logits bit-identical, loss {a['loss']:.6f}")


@pytest.mark.slow
def test_narrow_staged_epilogue_matches_quad(cuda, monkeypatch):
    """Round 4: the fp32 output tile of gemm_dma_narrow_kernel leaves through LDS as 16-byte row stores (narrow_store_staged: the hoisted
    encoder projections / ctx W_a / d X products and the narrow data gradients) instead of 64 four-byte stores per wave.  The values are
    the same values: against AOCR_NO_NARROW_STAGED=1 (the quad epilogue) context, logits and loss must be BIT-identical.  B = 128, W = 260
    (T = 64): M = B T = 8192 rows, the smallest shape at He = 256 whose hoisted products take that kernel (AOCR_FORCE_DMA=1 for the convs)."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    monkeypatch.setenv("AOCR_FORCE_DMA", "1")
    out = {}
    for off in ("1", "0"):
        monkeypatch.setenv("AOCR_NO_NARROW_STAGED", off)
        m, O, ocfg, P, st, batch = make(cfg, B=128, W=260, maxlen=6, compute="bf16", max_decoder_l=8, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, ctx=m.get_tensor("context").clone(), logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(),
                        dfeats=m.get_tensor("dfeats").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    assert torch.equal(a["ctx"], b["ctx"]) and torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    assert torch.equal(a["dfeats"], b["dfeats"])
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):          # bias in front of a BatchNorm: exact gradient 0, only rounding noise
            continue
        assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k          # split-K atomics: summation order only
    print(f"[parity] staged vs quad epilogue of the narrow LDS-DMA kernel: context / logits / d(feats) bit-identical, loss {a['loss']:.5f}")


@pytest.mark.parametrize("B,W", [(8, 256), (3, 128), (2, 384), (5, 100), (3, 200), (2, 232)])
def test_halo_wgrad_kernel_matches_tap_tiled_kernel(cuda, monkeypatch, B, W):
    """Round 4: conv_wgrad_halo_kernel (filter gradient with N tiles of nine taps x 32 input channels, the input map's halo staged once per
    32-pixel row segment) against the one-tap-per-tile kernels it replaces (AOCR_NO_WGRAD_HALO=1) on the same bf16 operands.  Only the fp32
    summation order over pixels differs (different split-K ranges), so every conv filter gradient must agree to accumulation noise; the
    forward pass and every other gradient do not involve the kernel and must be bit-identical.  W = 256 / 128 / 384 -> feature-map rows of
    64 / 32 / 96 pixels (2 / 1 / 3 segments per row; image borders inside and between the segments); AOCR_FORCE_DMA=1 selects it at these
    batch sizes.  Round 6, ragged rows: W = 100 / 200 / 232 -> rows of 25 and 50 / 50 and 100 / 58 and 116 pixels (the reference-default shape's maps: one or
    two segments per row, the last one 25 / 18 / 4 / 26 / 20 pixels with zero slots behind them)."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    monkeypatch.setenv("AOCR_FORCE_DMA", "1")
    out = {}
    for off in ("1", "0"):
        if off == "1": monkeypatch.setenv("AOCR_NO_WGRAD_HALO", "1")
        else: monkeypatch.delenv("AOCR_NO_WGRAD_HALO", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    worst = ("", 0.0)
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        e = relerr(b["grads"][k], a["grads"][k])
        if e > worst[1]: worst = (k, e)
        assert e < (2e-5 if k.startswith("cnn.conv") and k.endswith(".w") else 1e-6), (k, e)
    print(f"[parity] halo-resident filter gradient vs tap-tiled kernels, B={B} W={W}: worst relative difference {worst[1]:.2e} ({worst[0]})")


@pytest.mark.slow
def test_encoder_dx_single_product_matches_two_products(cuda, monkeypatch):
    """Round 4: the encoder's input gradient d X = d z_fw W_i2h_fw + d z_bw W_i2h_bw (model.lua:675 copy, :689 add) as ONE product over the
    concatenated K range (gemm_hh_cat, LoadKhCat) against the two products it replaces (AOCR_NO_HH_CAT=1: the second adds to the first's
    output).  Same bf16 operands; only the fp32 summation order differs (one accumulator over 2 x 4He instead of two rounded sums), so
    d(feats) agrees to accumulation noise, and everything upstream of it (forward pass, decoder / encoder recurrent gradients) is
    bit-identical.  B = 256, W = 204 (T = 50): the smallest C3-like shape whose B T x 512 product fills the narrow LDS-DMA kernel's grid."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for off in ("1", "0"):
        monkeypatch.setenv("AOCR_NO_HH_CAT", off)
        m, O, ocfg, P, st, batch = make(cfg, B=256, W=204, maxlen=6, compute="bf16", max_decoder_l=8, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, dfeats=m.get_tensor("dfeats").clone(), dctx=m.get_tensor("dcontext").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out["0"]
    assert a["loss"] == b["loss"] and torch.equal(a["dctx"], b["dctx"])
    e = relerr(b["dfeats"], a["dfeats"])
    print(f"[parity] d X as one product over [d z_fw | d z_bw]: d(feats) relative difference {e:.2e} against the two-product form")
    assert 0 < e < 1e-5                                      # (> 0: the single-product path really ran)
    for k in ("enc_fw.l1.i2h.w", "enc_bw.l1.h2h.w", "dec.attn.wa"):
        assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k
    for k in ("cnn.conv7.w", "cnn.conv6.w", "cnn.conv2.w"):
        assert cosine(b["grads"][k], a["grads"][k]) > 0.9999, k


@pytest.mark.slow
@pytest.mark.parametrize("B,W", [(8, 256), (16, 128)])
def test_bf16_data_gradient_maps_match_fp32_maps(cuda, monkeypatch, B, W):
    """Round 4, OPT-IN path (AOCR_DX16=1; measured -0.055 ms per C3 step, NOT the default): the data gradients of conv4-conv7 leave their
    kernels as bf16 (conv_backward_data's dx16: the staged 256 x 256 tile) instead of fp32 -- their only readers, the BatchNorm backward and
    un-pool passes, round their own output to bf16 for the next contraction anyway.  Against the default (fp32 maps): forward pass, loss and
    d(feats) bit-identical; the CNN gradients agree to one bf16 rounding of a gradient map (no ReLU / arg-max decision depends on a gradient,
    so nothing flips): cosine >= 0.9999, max-norm <= 5e-2.  It is off by default because that rounding is outside the "bf16 contraction
    operands" model the oracle tests hold the product to (tests/test_configs_gpu.py: lowest cosine 0.99998 -> 0.99983 with it on).
    AOCR_FORCE_DMA=1 selects the 256 x 256 kernels at these batch sizes."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    monkeypatch.setenv("AOCR_FORCE_DMA", "1")
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("AOCR_DX16", on)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[on] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dfeats=m.get_tensor("dfeats").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["0"], out["1"]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"] and torch.equal(a["dfeats"], b["dfeats"])
    worst = ("", 1.0, 0.0); differs = False
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        if not k.startswith("cnn."):
            assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k          # (grouped weight gradients: split-K atomics, summation order only)
            continue
        c, e = cosine(b["grads"][k], a["grads"][k]), relerr(b["grads"][k], a["grads"][k])
        differs = differs or e > 1e-4
        if c < worst[1]: worst = (k, c, e)
        assert c > 0.9999 and e < 5e-2, (k, c, e)
    assert differs, "the bf16 maps were not taken"
    print(f"[parity] bf16 data-gradient maps (opt-in) vs fp32 maps, B={B} W={W}: lowest cosine {worst[1]:.7f} ({worst[0]}, max-norm {worst[2]:.2e})")


@pytest.mark.slow
@pytest.mark.parametrize("B,W", [(8, 256), (16, 128)])
def test_bn_backward_sums_from_dgrad_epilogue(cuda, monkeypatch, B, W):
    """Round 4, OPT-IN path (AOCR_BNB_FUSE=1; measured SLOWER at C3 -- BatchNorm 0.50 -> 0.365 ms but data gradients 0.767 -> 0.931 ms per step --
    and therefore not the default; kept as a tested negative result): the sums pass of the BatchNorm backward of conv3 / conv5 -- (sum d,
    sum d xhat) per channel -- comes from the staged fp32 tile of the data gradient that produces d A (conv4 / conv6: EpStore::bnb_part,
    tile256_store_f32) instead of a pass of its own over d A, x and the mask.  Against the default: the same values summed in a different order (fp32 over a thread's 32-64 rows, fp64 from
    there, instead of fp64 throughout): forward pass, loss and d(feats) bit-identical, BatchNorm weight / bias gradients and everything
    below them within accumulation noise.  AOCR_FORCE_DMA=1 selects the 256 x 256 kernels at these batch sizes."""
    cfg = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
    monkeypatch.setenv("AOCR_FORCE_DMA", "1")
    out = {}
    for on in ("0", "1"):
        monkeypatch.setenv("AOCR_BNB_FUSE", on)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
        loss = m.train_forward_backward(batch)
        out[on] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dfeats=m.get_tensor("dfeats").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["0"], out["1"]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"] and torch.equal(a["dfeats"], b["dfeats"])
    worst = ("", 0.0); differs = False
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        e = relerr(b["grads"][k], a["grads"][k])
        differs = differs or (k in ("cnn.bn3.w", "cnn.bn5.w") and e > 0)
        if e > worst[1]: worst = (k, e)
        # the bf16 gradient maps below a BatchNorm re-round when its sums move in the last bits: bf16 one-ulp flips of single map entries, far below the oracle tolerances
        assert e < (5e-3 if k.startswith("cnn.") else 1e-5), (k, e)
        if k.startswith("cnn."): assert cosine(b["grads"][k], a["grads"][k]) > 0.999999, k
    assert differs, "the fused sums were not taken"
    print(f"[parity] BatchNorm backward sums from the data-gradient epilogue vs the separate pass, B={B} W={W}: worst relative difference {worst[1]:.2e} ({worst[0]})")


@pytest.mark.slow
@pytest.mark.parametrize("B,W,tile", [(6, 72, ""), (64, 100, ""), (64, 100, "1"), (64, 100, "2"), (5, 200, "3")])
def test_f32t_kernel_matches_lds_f32_kernel(cuda, monkeypatch, B, W, tile):
    """Round 4: gemm_f32t_kernel (exact-fp32 conv forward / data gradient / nn.Linear products: branch-free staging, tile shape by grid size:
    128 x 128, 64 x 128, 64 x 64) against gemm_lds_f32_kernel (AOCR_NO_F32T=1).  Same LDS image and the same k order per output element, so
    feature maps, logits, loss and every gradient that does not pass through split-K atomics must be BIT-identical -- with the tile chosen
    by the dispatch ("" : (64, 100) is BASELINE configs[1], where the 3x3 layers take 64 x 64 tiles) and with each shape forced."""
    cfg = dict(enc_hidden=256 if B == 64 else 64, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for off in ("1", "0"):
        monkeypatch.setenv("AOCR_NO_F32T", off)
        if tile and off == "0":
            monkeypatch.setenv("AOCR_F32T_TILE", tile)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="f32")
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, feats=m.get_tensor("feats").clone(), conv6=m.get_tensor("conv6").clone(), dfeats=m.get_tensor("dfeats").clone(),
                        logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    monkeypatch.delenv("AOCR_F32T_TILE", raising=False)
    a, b = out["1"], out["0"]
    assert torch.equal(a["conv6"], b["conv6"]) and torch.equal(a["feats"], b["feats"]) and torch.equal(a["logits"], b["logits"])
    assert a["loss"] == b["loss"] and torch.equal(a["dfeats"], b["dfeats"])
    for k in a["grads"]:
        if k.endswith(".w") and (k.startswith("cnn.conv") or "lstm" in k or k.startswith("enc") or k.startswith("dec")):
            assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k            # split-K atomics: not bit-stable between two runs of one path
        elif k not in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):      # (a bias in front of a BatchNorm: exact gradient 0, rounding noise only)
            assert relerr(b["grads"][k], a["grads"][k]) < 1e-5, k            # (max-norm relative: an element-wise bound with atol 1e-7 can trip on an element that is ~0 by cancellation, whose split-K atomics arrive in another order)
    print(f"[parity] f32t (tile '{tile or 'auto'}') vs gemm_lds_f32_kernel, B={B} W={W}: conv6 / feats / logits / dfeats bit-identical, loss {a['loss']:.6f}")


@pytest.mark.parametrize("B,W", [(6, 72), (64, 100), (256, 256)])
def test_embedding_token_table_matches_tensor_path(cuda, monkeypatch, B, W):
    """Round 4: with the whole-sequence decoder kernels the embedding part of the first layer's gate input is read from the per-token table
    [V][4 Hd] (lookup W_i2h[:, :E]^T + biases) and the backward pass sums d z by token (segsum_by_token) instead of gathering a (L B, E)
    tensor, multiplying it, and scattering its gradient back.  Against AOCR_NO_EMB_TABLE=1: the forward values are the same dot products --
    logits and loss BIT-identical; the three gradients it touches (lookup, the embedding columns of W_i2h, both biases of layer 1) are sums
    in a different order and with the V-sized products in exact fp32 instead of bf16 operands."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for off in ("1", "0"):
        if off == "1":
            monkeypatch.setenv("AOCR_NO_EMB_TABLE", "1")
        else:
            monkeypatch.delenv("AOCR_NO_EMB_TABLE", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=9, compute="bf16", max_decoder_l=10, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        assert m.cluster_status() == 0
        m.shutdown()
    a, b = out["1"], out["0"]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    worst = 0.0
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue                                             # bias in front of a BatchNorm: exact gradient 0, rounding noise only
        e = relerr(b["grads"][k], a["grads"][k]); worst = max(worst, e)
        # lookup / w_i embedding columns / layer-1 biases: bf16-operand products replaced by exact fp32 sums; everything else: the same kernels (split-K atomics only)
        assert e < (6e-3 if ("lookup" in k or k.startswith("dec.l1.")) else 1e-4), (k, e)
    print(f"[parity] embedding token table vs tensor path, B={B} W={W}: logits / loss bit-identical, worst gradient difference {worst:.2e}")


@pytest.mark.slow
def test_grouped_dma_weight_gradients_match_transposed_read_kernel(cuda, monkeypatch):
    """Round 4: the hoisted recurrent weight gradients (dW = dz^T x over all time steps) with full 256 x 256 tiles and K >= 2048 run on
    wgrad_dma_grouped_kernel (LDS-DMA ring, slabs per k range + wgrad_slab_reduce_kernel), the encoder's on the side stream beside the
    decoder's.  Against AOCR_NO_WGRAD_DMA_GROUPED=1 + AOCR_NO_ENC_WGRAD_SIDE=1 (wgrad_tr_grouped_kernel, 128 x 128 tiles, atomics): the
    same bf16 products summed in a different order -- every gradient within 2e-5 of the other path's, everything upstream bit-identical.
    B = 128, W = 260: K = L B = 1280 ... the decoder problems need K >= 2048, so L = 17 (max_decoder_l 18) and the encoder's K = T B = 8192."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for off in ("1", "0"):
        if off == "1":
            monkeypatch.setenv("AOCR_NO_WGRAD_DMA_GROUPED", "1"); monkeypatch.setenv("AOCR_NO_ENC_WGRAD_SIDE", "1")
        else:
            monkeypatch.delenv("AOCR_NO_WGRAD_DMA_GROUPED", raising=False); monkeypatch.delenv("AOCR_NO_ENC_WGRAD_SIDE", raising=False)
        m, O, ocfg, P, st, batch = make(cfg, B=128, W=260, maxlen=16, compute="bf16", max_decoder_l=18, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[off] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        assert m.cluster_status() == 0
        m.shutdown()
    a, b = out["1"], out["0"]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    worst = 0.0
    for k in a["grads"]:
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):
            continue
        e = relerr(b["grads"][k], a["grads"][k]); worst = max(worst, e)
        assert e < 2e-5, (k, e)
    print(f"[parity] grouped LDS-DMA weight gradients vs transposed-read kernel: worst gradient difference {worst:.2e}")


@pytest.mark.parametrize("switch", ["AOCR_NO_CNN_WGRAD_SIDE", "AOCR_NO_SIDE_PROLOGUE", "AOCR_NO_SIDE2",
                                    "AOCR_NO_Q_SIDE", "AOCR_SIDE_GO_LATE", "AOCR_NO_SHADOW_SPLIT", "AOCR_ENC_DC_COPY", "AOCR_CONV1_RECOMPUTE",      # round 5's re-ordered streams (they had no A/B test)
                                    "AOCR_JOIN_BEFORE_CNN_BWD"])                                                                # round 6: no join between the hoisted recurrent gradients and the CNN backward pass
def test_side_stream_overlaps_match_in_line_order(cuda, monkeypatch, switch):
    """the round-4 stream-level overlaps -- CNN filter gradients on the side stream over double-buffered gradient maps, the step prologue
    (gradient zeroing, weight shadows, token table) beside conv1, the second side stream -- each against the same steps with the switch that puts the
    work back in line.  Three optimisation steps on one batch: a missing event dependency (a kernel reading a buffer the other stream has not finished, or
    has already overwritten) shows up as a different gradient vector / different parameters; what may differ is the order in which split-K partial sums meet."""
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for knob in ("", "1"):
        monkeypatch.delenv(switch, raising=False)
        if knob:
            monkeypatch.setenv(switch, knob)
        m, O, ocfg, P, st, batch = make(cfg, B=32, W=256, maxlen=11, compute="bf16", max_decoder_l=12, max_beam=1)
        images, targets, targets_eval = m._upload(batch)
        # Three steps at learning rate 0 (the parameters stay put, everything else -- gradient zeroing, shadows, double-buffered maps, the update kernel --
        # runs): every step must reproduce the same gradient vector.  With a real learning rate the run-to-run noise of step 0 (2.7e-8: the order in
        # which split-K partial sums meet) grows to 4e-3 .. 6e-2 by step 2 with or without a switch (bf16 operand rounding, ReLU / arg-max decisions,
        # learning rate 0.1 on a random-init model): that comparison says nothing.  A fourth step at learning rate 0.1 then compares the update.
        m.optim_state["learningRate"] = 0.0
        grads = []
        for i in range(3):
            m.train_step_device(images, targets, targets_eval, 32)
            grads.append(m.grad_params.clone())
        m.optim_state["learningRate"] = 0.1
        m.train_step_device(images, targets, targets_eval, 32)
        torch.cuda.synchronize()
        assert m.cluster_status() == 0
        out[knob] = (grads, m.params.clone(), m.bn_state.clone())
        m.shutdown()
    (g0, p0, b0), (g1, p1, b1) = out[""], out["1"]
    for i, (a, b) in enumerate(zip(g0, g1)):
        rel = ((a - b).norm() / b.norm()).item()
        own = ((a - g0[0]).norm() / g0[0].norm()).item()
        assert rel < 1e-5 and own < 1e-5, (switch, i, rel, own)
    assert (p0 - p1).abs().max().item() < 1e-5 and (b0 - b1).abs().max().item() < 1e-5

