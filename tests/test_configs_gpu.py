"""Every BASELINE.json configuration and the reference's own default shape as -m gpu tests:

* reference default (src/train.lua:41,47): He = 512, Hd = 1024, Ld = 2 -- small batch against the fp64 oracle (logits 1e-4 in
  fp32 mode, every gradient tensor), bf16 production dispatch against the oracle, and a full-size (B = 400) property run;
* configs[2] C3 (32x256, B = 256, bf16): ALL gradient tensors against the fp64 oracle's hand-ordered BPTT, with stated
  relative / cosine tolerances;
* configs[3] C4 (W in 64..800 at He = 256, 64 rows per GPU): against the oracle where the fp64 CPU run is short, the
  size-independent properties (linearity in d(loss), batch-permutation equivariance) at W = 800;
* configs[4] C5 (128x1024 strips, 2-layer BiLSTM(512), beam 5): full geometry, properties with assertions.

The oracle (oracle/oracle_torch.py) is the checker only; every product call goes through the C ABI of libaocr."""
import numpy as np
import pytest
import torch

from test_step_gpu import make, relerr, _gpu_cnn_decisions
from tol import check_logits

pytestmark = pytest.mark.gpu

NOISY = ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b")     # bias in front of a BatchNorm: exact gradient 0, rounding noise only
REF_DEFAULT = dict(enc_hidden=512, enc_layers=1, dec_layers=2, input_feed=True)     # train.lua:47-49 + README.md:4 (-input_feed)
C3 = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)


def cosine(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def tensors(batch):
    return tuple(torch.from_numpy(np.asarray(x)) for x in batch[:3])


def check_bf16_gradients(tag, grads, G, Gq):
    """bf16 mode against BOTH oracles.  G: the plain fp64 restatement.  Gq: the same restatement with both operands of every
    contraction rounded to bf16 (oracle_torch.operand_rounding: what the product's bf16 mode does; fp64 accumulate, fp64 backward).
    Against Gq the ReLU masks and pool arg-max routes are those of a rounded forward pass: the recurrent stacks and everything from
    BatchNorm 7 upwards are held to cosine >= 0.9995 (measured >= 0.99996).  Below BatchNorm 7 two bf16 forward passes that differ
    only in accumulation order (fp32 MFMA here, fp64 in the oracle) still drift apart: a 1e-5 difference of a sum flips the bf16
    rounding of ~1 % of a layer's activations by one ulp (0.4 %), six conv layers later the pre-activations differ by ~1e-3 of their
    spread, and ~0.1 % of the ReLU / max-pool decisions flip (tools/diag_bf16_grad.py: 515 of 507 904 entries of d(conv7 output)
    differ, each by a whole term).  Every flip re-routes one gradient term, so the conv stack is held to cosine >= 0.995 against Gq
    (measured 0.9967-0.9989) and >= 0.96 against G (measured 0.967-0.994), where ~1 % of the decisions differ."""
    bad = []
    for k, g in G.items():
        if k in NOISY:
            continue
        r, c = relerr(grads[k], g), cosine(grads[k], g)
        rq, cq = relerr(grads[k], Gq[k]), cosine(grads[k], Gq[k])
        print(f"[parity] {tag} grad {k:22s} vs fp64: rel {r:.3e} cos {c:.6f} | vs bf16-operand oracle: rel {rq:.3e} cos {cq:.6f}")
        early = k.startswith("cnn.") and not k.startswith("cnn.bn7")
        if c < (0.96 if early else 0.999) or cq < (0.995 if early else 0.9995):
            bad.append((k, r, c, rq, cq))
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ reference default, He = 512
def test_reference_default_he512_fp32_vs_oracle(cuda):
    """src/train.lua:47 -encoder_num_hidden 512 (Hd = 1024), 32x100 crops, exact-fp32 MFMA mode, B = 6: logits 1e-4, loss,
    every gradient tensor against the fp64 oracle's hand-ordered BPTT."""
    m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=6, W=100, maxlen=9, max_decoder_l=12, max_beam=1)
    img, tgt, tge = tensors(batch)
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    print(f"[parity] He=512 fp32: logits max-abs {e:.3e}; loss {loss:.5f} vs {float(loss_ref) * 6:.5f}")
    check_logits(lg, aux["logits"], "f32", "He=512")
    assert abs(loss - float(loss_ref) * 6) < 1e-4 * abs(loss)
    grads = m.get_gradients()
    worst = ("", 0.0)
    for k, g in G.items():
        e = relerr(grads[k], g) if g.abs().max() > 1e-9 else (grads[k].double() - g).abs().max().item()
        if e > worst[1]: worst = (k, e)
        assert e < 2e-3, (k, e)
    print(f"[parity] He=512 fp32: {len(G)} gradient tensors, worst rel {worst[1]:.3e} ({worst[0]})")
    m.shutdown()


@pytest.mark.parametrize("B,big", [(16, False), (48, False), (48, True), (48, "stepl"), (70, "stepl")])
def test_reference_default_he512_bf16_vs_oracle(cuda, monkeypatch, B, big):
    """The same shape through the production bf16 dispatch (whole-sequence encoder kernels at He = 512, B % 16 == 0).
    big: the large-batch route of the decoder's step products (round 5: 128 x 128 LDS-DMA tiles over [x0 | x1] x [W0 | W1] + an elementwise cell pass --
    an opt-in route, AOCR_BIG_STEP=1: measured slower than the step kernels at the reference's default batch of 400, see ops_gemm.hip) forced at this batch, against the same oracle bounds.
    "stepl": the round-6 large-batch step kernels (stepl.h: LDS-DMA ring, eight waves, four-unit gate epilogue; default from ~320 rows at Hd = 1024) forced at this
    batch (B = 70: a ragged second row block), against the same oracle bounds."""
    if big == "stepl":
        monkeypatch.setenv("AOCR_STEPL_MIN_WGS", "1")
    elif big:
        monkeypatch.setenv("AOCR_BIG_STEP", "1"); monkeypatch.setenv("AOCR_BIG_STEP_MIN_ROWS", "16")
    m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=B, W=100, maxlen=9, compute="bf16", max_decoder_l=12, max_beam=1)
    img, tgt, tge = tensors(batch)
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    ec = (m.get_tensor("context").double() - aux["context"]).abs().max().item()
    print(f"[parity] He=512 bf16 B={B}: context max-abs {ec:.3e}, logits max-abs {e:.3e}; loss {loss:.4f} vs {float(loss_ref) * B:.4f}")
    check_logits(lg, aux["logits"], "bf16", f"He=512 B={B}")
    assert ec < 2e-2
    assert abs(loss - float(loss_ref) * B) < 5e-3 * abs(loss)
    grads = m.get_gradients()
    for k in ("proj.w", "dec.attn.wa", "dec.attn.wc", "dec.lookup", "dec.l1.i2h.w", "dec.l1.i2h.b", "dec.l2.h2h.w", "enc_fw.l1.h2h.w", "enc_bw.l1.h2h.w", "enc_fw.l1.i2h.w",
              "enc_bw.l1.i2h.b", "cnn.conv7.w", "cnn.bn7.w"):
        c = cosine(grads[k], G[k]); r = relerr(grads[k], G[k])
        print(f"[parity] He=512 bf16 B={B} grad {k:18s} rel {r:.3e} cosine {c:.6f}")
        # the CNN tensors sit behind BatchNorm over a tiny batch (16 x 24 positions): bf16 rounding of the activations is amplified
        # by 1/std of near-constant channels, so they are held to a looser direction bound than the recurrent stacks
        assert c > (0.98 if k.startswith("cnn.") else 0.995), (k, c)
    m.shutdown()


def test_reference_default_full_size_properties(cuda):
    """Reference default at full size: batch 400 (train.lua:41), He = 512, 32x100, L = 24, bf16: linearity of the backward pass in
    d(loss) and equivariance under a permutation of the batch (size-independent; the fp64 oracle would need minutes here)."""
    B = 400
    m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=B, W=100, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
    _properties(m, ocfg, batch, B, "bf16")
    m.shutdown()


@pytest.mark.parametrize("B", [400, 390])
def test_reference_default_step_kernel_routes_agree(cuda, monkeypatch, B):
    """The decoder launch chain of the reference-default shape (Hd = 1024: no whole-sequence decoder kernel) at full batch, three routes of its step products:
    the round-6 large-batch kernels (stepl.h, the default at this size), gemm_step_kernel with two row tiles per workgroup (round 5, AOCR_NO_STEPL=1) and with
    one (AOCR_NO_STEP_MT2=1).  The two gemm_step_kernel forms do the same arithmetic in the same order -- logits and loss BIT-identical (the
    two-tile form had no test at the batch that selects it; B = 390: its second row tile is partly, the last workgroup's wholly, past the end); stepl.h sums the
    K range in another order -- every tensor within summation-order noise of the chain."""
    out = {}
    for name, env in (("stepl", {}), ("mt2", {"AOCR_NO_STEPL": "1"}), ("mt1", {"AOCR_NO_STEPL": "1", "AOCR_NO_STEP_MT2": "1"})):
        for k in ("AOCR_NO_STEPL", "AOCR_NO_STEP_MT2"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=B, W=100, maxlen=9, compute="bf16", max_decoder_l=12, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[name] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        assert m.cluster_status() == 0
        m.shutdown()
    a, b, c = out["mt1"], out["mt2"], out["stepl"]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"], "two row tiles per workgroup changed the arithmetic"
    e = (c["logits"] - a["logits"]).abs().max().item()
    print(f"[parity] B={B} stepl vs step kernels: logits max-abs {e:.3e}, loss {c['loss']:.4f} vs {a['loss']:.4f}")
    assert e < 5e-3 and abs(c["loss"] - a["loss"]) < 1e-3 * abs(a["loss"])
    worst = ("", 0.0)
    for k in a["grads"]:
        if k in NOISY:
            continue
        r, cs = relerr(c["grads"][k], a["grads"][k]), cosine(c["grads"][k], a["grads"][k])
        if r > worst[1]: worst = (k, r)
        assert cs > (0.995 if k.startswith("cnn.") else 0.9999) and r < (0.3 if k.startswith("cnn.") else 3e-2), (k, r, cs)
    print(f"[parity] B={B} stepl vs step kernels: worst gradient rel {worst[1]:.3e} ({worst[0]})")


@pytest.mark.parametrize("B,W", [(48, 100), (21, 256)])
def test_reference_default_chain_scores_against_premultiplied_context(cuda, monkeypatch, B, W):
    """Round 6: the Hd = 1024 launch chain scores attention against bf16(ctx W_a) (LSTM.lua:131-137: ctx[t] . (W_a h) = (ctx W_a)[t] . h), as the whole-sequence
    kernels of Hd = 512 do: no q = W_a h launch per forward step, no K = Hd product in the second cell launch of the backward step (its attention part comes
    out of the attention backward kernel as a second weighted sum), q of all steps as one hoisted product for d(context).  Against AOCR_NO_CHAIN_CTXA=1 (q per
    step): the same number of bf16 roundings per score term in another place -- logits 2e-3, every gradient tensor as two summation orders
    (W = 256: T = 63, the 16-wave form of the kernel; W = 100: T = 24, the 8-wave form)."""
    out = {}
    for knob in ("", "1"):
        monkeypatch.delenv("AOCR_NO_CHAIN_CTXA", raising=False)
        if knob:
            monkeypatch.setenv("AOCR_NO_CHAIN_CTXA", knob)
        m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=B, W=W, maxlen=9, compute="bf16", max_decoder_l=12, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dctx=m.get_tensor("dcontext").clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out[""]
    e = (a["logits"] - b["logits"]).abs().max().item()
    ed = relerr(b["dctx"], a["dctx"])
    print(f"[parity] chain on ctx W_a vs q per step, B={B} W={W}: logits max-abs {e:.3e}, d(context) rel {ed:.3e}, loss {b['loss']:.4f} vs {a['loss']:.4f}")
    assert e < 2e-3 and ed < 3e-2 and abs(a["loss"] - b["loss"]) < 1e-3 * abs(a["loss"])
    for k in a["grads"]:
        if k in NOISY:
            continue
        r, cs = relerr(b["grads"][k], a["grads"][k]), cosine(b["grads"][k], a["grads"][k])
        assert cs > (0.99 if k.startswith("cnn.") else 0.9995) and r < (0.3 if k.startswith("cnn.") else 5e-2), (k, r, cs)


@pytest.mark.parametrize("cfg,B,W", [(C3, 128, 256), (C3, 100, 100), (REF_DEFAULT, 90, 100)])
def test_projector_criterion_one_launch(cuda, monkeypatch, cfg, B, W):
    """Round 6: a training step's projector (model.lua:594-612), criterion (criterion.lua:3-9) and projector data gradient (model.lua:648) run as one launch
    (project_loss_kernel) instead of three between the two decoder passes.  Each part keeps the arithmetic and the order of the launch it replaces, so against
    AOCR_NO_PROJ_FUSE=1: logits, loss, d logits and d out bit-equal -- and with them every gradient tensor the decoder BPTT computes from d out alone (those
    summed by split-K atomics differ by their run-to-run order).  B = 100 / 90: a ragged last 32-row block (rows = 1200 / 1080)."""
    out = {}
    for knob in ("", "1"):
        monkeypatch.delenv("AOCR_NO_PROJ_FUSE", raising=False)
        if knob:
            monkeypatch.setenv("AOCR_NO_PROJ_FUSE", knob)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=11, compute="bf16", max_decoder_l=12, max_beam=1)
        loss = m.train_forward_backward(batch)
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dlogits=m.get_tensor("dlogits").clone(), dout=m.get_tensor("dout_proj").clone(),
                         dctx=m.get_tensor("dcontext").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out[""]
    assert a["logits"].shape[0] * a["logits"].shape[1] >= 1024
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    assert torch.equal(a["dlogits"], b["dlogits"]) and torch.equal(a["dout"], b["dout"])
    worst = 0.0
    for k in a["grads"]:
        if k in NOISY:
            continue
        worst = max(worst, relerr(b["grads"][k], a["grads"][k]))
    print(f"[parity] projector + criterion + d out in one launch vs three: logits / loss / d logits / d out bit-equal; worst gradient tensor rel {worst:.2e} (atomic split-K order)")
    assert worst < 2e-5 and relerr(b["dctx"], a["dctx"]) < 2e-5


def test_streamed_attention_backward_one_pass(cuda, monkeypatch):
    """Round 6: above 128 context rows (Hd = 1024) the attention kernels stream the context instead of holding it in registers; the backward form made two
    passes (d a_t = ctx_t . d c and dot = sum a_t d a_t first, then d q = sum a_t (d a_t - dot) ctx_t).  It is one pass now: d q = sum_t a_t d a_t ctx_t -
    dot c_fwd with c_fwd the forward pass's saved weighted context (LSTM.lua:131-145's MixtureTable output).  Against AOCR_ATTN_BWD_TWO_PASS=1 on the
    reference-default model at W = 560 (T = 139): the forward pass is the same code (logits bit-equal), gradients as two summation orders."""
    out = {}
    for knob in ("", "1"):
        monkeypatch.delenv("AOCR_ATTN_BWD_TWO_PASS", raising=False)
        if knob:
            monkeypatch.setenv("AOCR_ATTN_BWD_TWO_PASS", knob)
        m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=6, W=560, maxlen=9, compute="bf16", max_decoder_l=12, max_beam=1)
        loss = m.train_forward_backward(batch)
        assert m.get_tensor("context").shape[1] == 139
        out[knob] = dict(loss=loss, logits=m.get_tensor("logits")[:, :, :ocfg.vocab].clone(), dctx=m.get_tensor("dcontext").clone(),
                         grads={k: v.clone() for k, v in m.get_gradients().items()})
        m.shutdown()
    a, b = out["1"], out[""]
    assert torch.equal(a["logits"], b["logits"]) and a["loss"] == b["loss"]
    ed = relerr(b["dctx"], a["dctx"])
    worst = ("", 0.0)
    for k in a["grads"]:
        r, cs = relerr(b["grads"][k], a["grads"][k]), cosine(b["grads"][k], a["grads"][k])
        if r > worst[1]:
            worst = (k, r)
        if k in NOISY:
            continue
        assert cs > 0.9999 and r < 1e-2, (k, r, cs)
    print(f"[parity] streamed attention backward, one pass vs two: d(context) rel {ed:.3e}, worst gradient tensor {worst[0]} rel {worst[1]:.3e}")
    assert ed < 1e-3


@pytest.mark.parametrize("beam,B,force", [(1, 21, False), (3, 21, True), (5, 6, False)])
def test_reference_default_decode_chain_on_shadows(cuda, monkeypatch, beam, B, force):
    """Round 6: the decode launch chain (Hd = 1024 has no whole-sequence decoder: `-phase test` of the reference-default model, BASELINE config 5's beam-5 decode)
    keeps bf16 shadows of its beam state -- gate / attention epilogues write them, the gather by parent converts -- so its step products are the training chain's
    (from ~130 rows stepl.h; scores against ctx W_a) instead of the fp32-activation kernels: 121.7 -> 64.7 us per step at 400 rows.  Both forms round the same
    fp32 values to bf16, so against AOCR_NO_DECODE_SHADOWS=1: labels equal (a near-tie may move one row), scores / gold scores / loss to bf16-path noise; and
    against the fp64 oracle's decode_beam.  force: stepl.h at this small batch (AOCR_STEPL_MIN_WGS=1)."""
    out = {}
    if force:
        monkeypatch.setenv("AOCR_STEPL_MIN_WGS", "1")
    for knob in ("", "1"):
        monkeypatch.delenv("AOCR_NO_DECODE_SHADOWS", raising=False)
        if knob:
            monkeypatch.setenv("AOCR_NO_DECODE_SHADOWS", knob)
        m, O, ocfg, P, st, batch = make(REF_DEFAULT, B=B, W=100, maxlen=7, compute="bf16", max_decoder_l=12, max_beam=5)
        st = {k: (v + 0.05 if k.endswith("rm") else v * 1.3) for k, v in st.items()}
        m.set_parameters(P, st)
        loss, stats = m.step(batch, True, beam)
        out[knob] = (loss, m._dec_out.labels.copy(), m._dec_out.scores.copy(), m._dec_out.gold_scores.copy())
        if not knob and B <= 8:
            img, tgt, tge = tensors(batch)
            ref = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=12)
            same = (m._dec_out.labels == ref["labels"].numpy().astype(np.int32)).all(axis=1)
            print(f"[parity] decode chain on shadows, beam {beam} B={B}: {int(same.sum())}/{B} label rows equal to the oracle's; loss {loss:.4f} vs {float(ref['loss']):.4f}")
            assert same.sum() >= B - 1 and abs(loss - float(ref["loss"])) < 5e-3 * max(1.0, float(ref["loss"]))
            assert np.abs(m._dec_out.gold_scores - ref["gold_scores"].numpy()).max() < 5e-2
        m.shutdown()
    (l1, lab1, sc1, g1), (l0, lab0, sc0, g0) = out[""], out["1"]
    same = (lab1 == lab0).all(axis=1)
    print(f"[parity] decode chain on shadows vs fp32-activation kernels, beam {beam} B={B}: {int(same.sum())}/{B} label rows equal; loss {l1:.4f} vs {l0:.4f}")
    assert same.sum() >= B - 1 and abs(l1 - l0) < 2e-3 * abs(l0)
    assert np.abs(sc1 - sc0)[same].max() < 2e-2 and np.abs(g1 - g0).max() < 2e-2


def _properties(m, ocfg, batch, B, compute, lin_tol=None, perm_tol=None):
    img, tgt, tge, nnz, names = batch
    loss1 = m.train_forward_backward(batch, grad_scale=1.0 / B)
    assert np.isfinite(loss1) and loss1 > 0
    g1 = {k: v.clone() for k, v in m.get_gradients().items()}
    assert all(torch.isfinite(v).all() for v in g1.values())
    lg1 = m.get_tensor("logits")[:, :, :ocfg.vocab].clone()
    loss4 = m.train_forward_backward(batch, grad_scale=4.0 / B)
    g4 = m.get_gradients()
    assert loss4 == pytest.approx(loss1, rel=1e-6)
    worst = max(relerr(g4[k], 4 * g1[k]) for k in g1 if k not in NOISY)
    print(f"[property] {compute} B={B}: linearity in d(loss): worst rel {worst:.2e}")
    assert worst < (lin_tol or (2e-3 if compute == "bf16" else 2e-4))
    perm = np.random.default_rng(0).permutation(B)
    pbatch = [np.asarray(img)[perm], np.asarray(tgt)[perm], np.asarray(tge)[perm], nnz, [names[i] for i in perm]]
    lossp = m.train_forward_backward(pbatch, grad_scale=1.0 / B)
    gp = m.get_gradients()
    lgp = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lgp - lg1[:, perm]).abs().max().item()
    worst = max(relerr(gp[k], g1[k]) for k in g1 if k not in NOISY)
    print(f"[property] {compute} B={B}: batch permutation: logits max-abs {e:.2e}, loss {lossp:.3f} vs {loss1:.3f}, worst gradient rel {worst:.2e}")
    assert e < (2e-2 if compute == "bf16" else 2e-5)
    assert lossp == pytest.approx(loss1, rel=(1e-4 if compute == "bf16" else 1e-6))
    assert worst < (perm_tol or (3e-2 if compute == "bf16" else 1e-3))


# ------------------------------------------------------------------------------------------------ C3: bf16 gradients vs the oracle
def test_c3_bf16_all_gradients_vs_oracle(cuda):
    """BASELINE configs[2] at full size (32x256, B = 256, He = 256, L = 24) through the production bf16 dispatch: loss and ALL
    gradient tensors against the fp64 oracle AND against the same oracle with bf16-rounded operands (check_bf16_gradients)."""
    B = 256
    m, O, ocfg, P, st, batch = make(C3, B=B, W=256, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
    img, tgt, tge = tensors(batch)
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    with O.operand_rounding("bf16"):
        loss_q, Gq, rq, _ = O.train_step_autograd(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e, eq = (lg.double() - aux["logits"]).abs().max().item(), (lg.double() - rq["logits"].detach()).abs().max().item()
    print(f"[parity] C3 bf16: loss {loss:.4f} vs fp64 {float(loss_ref) * B:.4f} vs bf16-operand oracle {float(loss_q) * B:.4f}; logits max-abs {e:.3e} / {eq:.3e}")
    assert abs(loss - float(loss_ref) * B) < 2e-3 * abs(loss) and abs(loss - float(loss_q) * B) < 2e-4 * abs(loss)
    check_logits(lg, aux["logits"], "bf16", "C3")
    assert eq < 1e-3                                     # against the bf16-operand oracle: measured 2.6e-4
    grads = m.get_gradients()
    check_bf16_gradients("C3 bf16", grads, G, Gq)
    # The conv stack's cosine of 0.97-0.999 is ReLU / arg-max DECISION flips between two bf16 forward passes (check_bf16_gradients):
    # tested here by imposing the GPU's own decisions on the bf16-operand oracle -- then every tensor agrees to bf16 arithmetic noise.
    dec = _gpu_cnn_decisions(m, B)
    with O.operand_rounding("bf16"):
        loss_d, Gd, _, _ = O.train_step_autograd(P, st, ocfg, img, tgt, tge, cnn_decisions=dec)
    worst = ("", 1.0)
    for k, g in Gd.items():
        if k in NOISY:
            continue
        c, r = cosine(grads[k], g), relerr(grads[k], g)
        print(f"[parity] C3 bf16, GPU decisions imposed on the bf16-operand oracle: {k:22s} rel {r:.3e} cos {c:.6f}")
        if c < worst[1]: worst = (k, c)
        assert c > 0.9999 and r < 2e-2, (k, c, r)                   # measured: cosine >= 0.99998, rel <= 6e-3 (conv1 included)
    print(f"[parity] C3 bf16 with imposed decisions: worst cosine {worst[1]:.6f} ({worst[0]})")
    m.shutdown()


# ------------------------------------------------------------------------------------------------ C4: variable widths, 64 rows per GPU
@pytest.mark.parametrize("W,compute", [(64, "f32"), (64, "bf16"), (416, "bf16")])
def test_c4_width_vs_oracle(cuda, W, compute):
    """BASELINE configs[3]: one width bucket per step, 64 rows per GPU, He = 256.  Against the fp64 oracle (forward + hand-ordered
    BPTT) at the widths whose CPU run is short: W = 64 (T = 15) and W = 416 (T = 103: generic-T attention, T > 64)."""
    B = 64
    m, O, ocfg, P, st, batch = make(C3, B=B, W=W, maxlen=23, compute=compute, max_decoder_l=24, max_beam=1)
    img, tgt, tge = tensors(batch)
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    print(f"[parity] C4 W={W} {compute}: T={aux['context'].shape[1]} logits max-abs {e:.3e}; loss {loss:.4f} vs {float(loss_ref) * B:.4f}")
    check_logits(lg, aux["logits"], compute, f"C4 W={W}")
    assert abs(loss - float(loss_ref) * B) < (1e-4 if compute == "f32" else 3e-3) * abs(loss)
    grads = m.get_gradients()
    if compute == "bf16":
        with O.operand_rounding("bf16"):
            _, Gq, _, _ = O.train_step_autograd(P, st, ocfg, img, tgt, tge)
        check_bf16_gradients(f"C4 W={W} bf16", grads, G, Gq)
    else:
        for k, g in G.items():
            if k in NOISY:
                continue
            r, c = relerr(grads[k], g), cosine(grads[k], g)
            early = k.startswith("cnn.") and not k.startswith(("cnn.conv7", "cnn.bn7"))
            assert (c > 0.9995 and r < 5e-2) if early else r < 2e-3, (k, r, c)
    m.shutdown()


def test_c4_widest_bucket_properties(cuda):
    """BASELINE configs[3], widest bucket: W = 800 (T = 199), B = 64, He = 256, bf16 -- size-independent properties."""
    B = 64
    m, O, ocfg, P, st, batch = make(C3, B=B, W=800, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
    _properties(m, ocfg, batch, B, "bf16")
    m.shutdown()


def test_c4_bucket_sequence_one_model(cuda):
    """One model instance sized for the widest bucket steps through several width buckets in a row (what DataGen emits,
    data_gen.lua:92-120): every step's loss equals the loss of a fresh model fed that bucket alone."""
    import aocr
    B = 64
    m, O, ocfg, P, st, _ = make(C3, B=B, W=800, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
    for W in (800, 64, 416, 96):
        img, tgt, tge, nnz = O.synth_batch(B, W, max_len=23, min_len=4)
        batch = [img, tgt, tge, nnz, None]
        l_shared = m.train_forward_backward(batch)
        m2, _, _, _, _, _ = make(C3, B=B, W=W, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
        l_fresh = m2.train_forward_backward(batch)
        m2.shutdown()
        print(f"[property] C4 bucket W={W}: shared-workspace loss {l_shared:.4f} fresh {l_fresh:.4f}")
        assert l_shared == pytest.approx(l_fresh, rel=2e-4)
    m.shutdown()


# ------------------------------------------------------------------------------------------------ C5: 128x1024 strips
@pytest.mark.parametrize("B", [16, 256])
def test_c5_full_geometry(cuda, B):
    """BASELINE configs[4] at full geometry: 128x1024 strips (the CNN leaves 7 x 255 positions, T = 1785, row-major as View(512,-1)
    of cnn.lua:44 strings them), 2-layer BiLSTM(512), Hd = 1024, beam-width-5 decode; 16 strips per GPU (2 x 2 x 8 compute units of the encoder's
    groups busy) and 256 -- the batch bench.py reports the configuration at (round 6: BASELINE names none; at 256 the encoder's 16-row groups cover the
    chip: 2 directions x 16 groups x 8 CUs, and the decoder's step products run on stepl.h).  No oracle at this size (the structure is checked against it
    at 64x40 in test_step_gpu.py::test_tall_strips...): properties with assertions."""
    import aocr
    H, W, L = 128, 1024, 24
    m = aocr.Model().create(dict(encoder_num_hidden=512, encoder_num_layers=2, decoder_num_layers=2, input_feed=True, batch_size=B,
                                 img_h=H, max_img_w=W, max_decoder_l=30, max_beam=5, compute="bf16", learning_rate=0.1, seed=1))
    img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=5, max_len=L - 1, H=H)
    names = [str(i) for i in range(B)]
    batch = [img, tgt, tge, nnz, names]
    ocfg = type("C", (), {"vocab": 39})()
    _properties(m, ocfg, batch, B, "bf16", lin_tol=5e-3, perm_tol=6e-2)
    T = m.get_tensor("context").shape[1]
    assert T == 7 * 255
    # beam-5 decode: finite scores, equivariant under a batch permutation, gold loss = teacher-forced eval-mode loss
    loss, stats = m.step(batch, True, 5)
    out1 = m._dec_out
    assert np.isfinite(loss) and np.isfinite(out1.scores).all() and np.isfinite(out1.gold_scores).all()
    assert out1.labels.min() >= 1 and out1.labels.max() <= 39
    _, loss_tf = m.forward_logits(batch, training=False)
    assert loss == pytest.approx(loss_tf, rel=1e-3)
    perm = np.random.default_rng(1).permutation(B)
    pbatch = [np.asarray(img)[perm], np.asarray(tgt)[perm], np.asarray(tge)[perm], nnz, [names[i] for i in perm]]
    lossp, _ = m.step(pbatch, True, 5)
    out2 = m._dec_out
    same = (out2.labels == out1.labels[perm]).all(axis=1)
    print(f"[property] C5 beam 5: {int(same.sum())}/{B} label rows identical under permutation; loss {loss:.3f} vs {lossp:.3f}")
    assert same.sum() >= B - max(1, B // 64) and lossp == pytest.approx(loss, rel=1e-3)      # (a near-tie between two hypotheses may fall differently for a row)
    assert np.abs(out2.scores - out1.scores[perm])[same].max() < 2e-2
    # beam 5 never scores below greedy on the same model (the greedy path is inside the beam at every step unless pruned by
    # higher-scoring prefixes; the final answer is the max over the beam, model.lua:574)
    loss1, _ = m.step(batch, True, 1)
    g = m._dec_out
    frac = float((out1.scores >= g.scores - 1e-3).mean())
    print(f"[property] C5: beam-5 score >= greedy score on {frac * 100:.0f}% of rows")
    assert frac >= 0.9
    m.shutdown()
