"""Parity away from random init ("give the parity tests teeth").

At U(+-1/sqrt(fan_in)) every decoder logit is ~0.03, the attention is uniform over T and the loss is nnz * ln 39: an absolute
1e-4 bound on such logits is ~3e-3 relative, and a path that mis-scaled the attention scores would still pass.  Here the SAME
seeded weights are sharpened (oracle_torch.sharpen_params: W_a x80, projector x8, W_c x3, every LSTM matrix x3) so that
|logit| is O(1) (max 4-6), the attention peaks (mean entropy 0.3-0.7 nat) and the gates leave their linear range; and a
ten-step fp32 TRAJECTORY (feval + clipped SGD, two alternating batches) is followed against the oracle's own
train_step_manual + sgd_list, so that an error that compounds over updates (a wrong BatchNorm running statistic, a stale
shadow copy of a weight, a gradient that is right only at step 0) shows up in the parameters.

Reference rows: LSTM.lua:124-162 (attention), output_projector.lua:3-8, criterion.lua:3-9, model.lua:634-706, optim_sgd.lua:38-95."""
import numpy as np
import pytest
import torch

from test_step_gpu import make, relerr, cosine
from tol import check_logits

pytestmark = pytest.mark.gpu

NOISY = ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b")     # bias in front of a BatchNorm: exact gradient 0, rounding noise only
SMALL = dict(enc_hidden=32, enc_layers=1, dec_layers=2, input_feed=True)
C2 = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)


def tensors(batch):
    return tuple(torch.from_numpy(np.asarray(x)) for x in batch[:3])


def regime(O, P, st, ocfg, img, tgt, tge):
    with torch.no_grad():
        r = O.forward_train(P, {k: v.clone() for k, v in st.items()}, ocfg, img, tgt, tge, training=True)
    ent = O.attention_entropy(r)
    return float(ent.mean()), float(r["logits"].abs().max()), int(r["context"].shape[1])


@pytest.mark.parametrize("cfgkw,B,W,maxlen", [(SMALL, 5, 36, 6), (dict(enc_hidden=48, enc_layers=2, dec_layers=3, input_feed=True), 4, 52, 6),
                                              (dict(enc_hidden=32, enc_layers=1, dec_layers=2, input_feed=False), 5, 36, 6)])
def test_sharpened_fp32_small_all_gradients(cuda, cfgkw, B, W, maxlen):
    """fp32 mode, sharpened weights: logits within 1e-4 of the LARGEST logit (and 1e-4 absolute), loss, every gradient tensor <= 2e-3."""
    m, O, ocfg, P0, st, batch = make(cfgkw, B=B, W=W, maxlen=maxlen)
    P = O.sharpen_params(P0, **(dict(wa=200.0, proj=12.0) if cfgkw["enc_layers"] > 1 else {}))
    m.set_parameters(P, st)
    img, tgt, tge = tensors(batch)
    ent, top, T = regime(O, P, st, ocfg, img, tgt, tge)
    print(f"[sharp] He={ocfg.enc_hidden} Le={ocfg.enc_layers} feed={ocfg.input_feed}: T={T}, mean attention entropy {ent:.3f} nat (uniform {np.log(T):.3f}), max |logit| {top:.2f}")
    assert ent < 1.0 and top > 1.0
    loss_ref, G, aux, st_new = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e, _ = check_logits(lg, aux["logits"], "f32", "sharpened small")
    print(f"[sharp] logits max-abs {e:.3e} = {e / top:.2e} of the largest; loss {loss:.5f} vs {float(loss_ref) * B:.5f}")
    assert abs(loss - float(loss_ref) * B) < 1e-5 * abs(loss)
    ectx = (m.get_tensor("context").double() - aux["context"]).abs().max().item(); assert ectx < 1e-5, ectx
    grads = m.get_gradients()
    worst = ("", 0.0)
    for k, g in G.items():
        e = relerr(grads[k], g) if g.abs().max() > 1e-9 else (grads[k].double() - g).abs().max().item()
        if e > worst[1]: worst = (k, e)
        assert e < 2e-3, (k, e)
    print(f"[sharp] {len(G)} gradient tensors, worst rel {worst[1]:.3e} ({worst[0]})")
    m.shutdown()


@pytest.mark.parametrize("kw,strict", [(dict(lstm=2.0), True), (dict(), False)])
def test_sharpened_fp32_c2_shape(cuda, kw, strict):
    """BASELINE configs[1] (32x100, B = 64, He = 256, L = 24) with sharpened weights, exact-fp32 mode.
    strict (LSTM matrices x2): logits within 1e-4 of the LARGEST logit, every gradient from conv7 upwards <= 2e-3, the early convolutions to
    the cosine (ReLU / arg-max near-ties: DESIGN.md section 4).
    not strict (LSTM matrices x3, the setting of the small cases): 24 saturating recurrent steps with input feed amplify ANY fp32 rounding --
    the oracle's OWN float32 evaluation (same restatement, torch CPU float32: what an fp32 Torch7 run would give) then differs from its
    float64 evaluation by 1.9e-3 = 3e-4 of the largest logit -- so the yardstick is that number: the HIP path must be no further from the
    fp64 oracle than 4x the fp32 oracle is (measured 2.4x: different summation order, v_exp_f32 / v_rcp_f32 activations)."""
    B = 64
    m, O, ocfg, P0, st, batch = make(C2, B=B, W=100, maxlen=23, max_decoder_l=24, max_beam=1)
    P = O.sharpen_params(P0, **kw)
    m.set_parameters(P, st)
    img, tgt, tge = tensors(batch)
    ent, top, T = regime(O, P, st, ocfg, img, tgt, tge)
    with torch.no_grad():
        r32 = O.forward_train({k: v.float() for k, v in P.items()}, {k: v.float() for k, v in st.items()}, ocfg, img.float(), tgt, tge, training=True)
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    e32 = (r32["logits"].double() - aux["logits"]).abs().max().item()
    print(f"[sharp] C2 {kw}: T={T}, mean attention entropy {ent:.3f} nat (uniform {np.log(T):.3f}), max |logit| {top:.2f}; the oracle in float32 vs float64: {e32:.3e}")
    assert ent < 1.0 and top > 1.0
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    e = (lg.double() - aux["logits"]).abs().max().item()
    print(f"[sharp] C2 logits max-abs {e:.3e} = {e / top:.2e} of the largest = {e / e32:.2f} x the fp32 oracle's distance; loss {loss:.4f} vs {float(loss_ref) * B:.4f}")
    assert e < 4.0 * e32 + 1e-6, (e, e32)
    if strict:
        assert e < 1e-4 * top, (e, top)
    assert abs(loss - float(loss_ref) * B) < 2e-5 * abs(loss)
    grads = m.get_gradients()
    worst = ("", 0.0)
    for k, g in G.items():
        if k in NOISY:
            continue
        r, c = relerr(grads[k], g), cosine(grads[k], g)
        early = k.startswith("cnn.") and not k.startswith(("cnn.conv7", "cnn.bn7"))
        if not early and r > worst[1]: worst = (k, r)
        if strict:
            assert (c > 0.9995 and r < 5e-2) if early else r < 2e-3, (k, r, c)
        else:
            assert c > 0.999, (k, r, c)
    print(f"[sharp] C2 {kw}: every gradient from conv7 upwards within {worst[1]:.3e} ({worst[0]})")
    m.shutdown()


BF16_SHARP = dict(wa=12.0, proj=8.0, lstm=2.0, wc=2.0)


@pytest.mark.parametrize("B,W", [(32, 100), (64, 256)])
def test_sharpened_bf16_production_dispatch(cuda, B, W):
    """bf16 mode through the production dispatch (cluster kernels, B % 32 == 0, He = 256), sharpened weights.  Two references: the fp64
    oracle (bound relative to the largest logit: the absolute size of a bf16 error grows with the logits) and the same oracle with
    bf16-rounded operands -- the arithmetic the product implements.
    The sharpening is MILDER here than in the fp32 tests (W_a x12 instead of x80): an attention score is a 512-term product sum whose
    bf16 operand rounding error grows with the scale, and softmax turns a score error d into a factor e^d -- at x80 the fp64 oracle and
    the bf16-operand ORACLE already differ by 3-4 logit units (measured, round 4), i.e. there is no bf16 answer to hold the product to.
    At x12 the logits are O(1) and the attention is no longer uniform, and the two oracles still agree to a few percent."""
    m, O, ocfg, P0, st, batch = make(C2, B=B, W=W, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
    P = O.sharpen_params(P0, **BF16_SHARP)
    m.set_parameters(P, st)
    img, tgt, tge = tensors(batch)
    with torch.no_grad():
        r = O.forward_train(P, {k: v.clone() for k, v in st.items()}, ocfg, img, tgt, tge, training=True)
        with O.operand_rounding("bf16"):
            rq = O.forward_train(P, {k: v.clone() for k, v in st.items()}, ocfg, img, tgt, tge, training=True)
    ent, top = float(O.attention_entropy(r).mean()), float(r["logits"].abs().max())
    logits, loss = m.forward_logits(batch, training=True)
    e = (logits.double() - r["logits"]).abs().max().item()
    eq = (logits.double() - rq["logits"]).abs().max().item()
    eo = (rq["logits"] - r["logits"]).abs().max().item()
    print(f"[sharp] bf16 B={B} W={W}: entropy {ent:.3f} nat, max |logit| {top:.2f}; logits max-abs vs fp64 oracle {e:.3e} ({e / top:.2e} of the largest), "
          f"vs bf16-operand oracle {eq:.3e}; the two oracles differ by {eo:.3e}; loss {loss:.3f} vs {float(r['loss']) * B:.3f} / {float(rq['loss']) * B:.3f}")
    assert top > 1.0 and ent < 0.9 * np.log(r["context"].shape[1])      # O(1) logits, attention visibly away from uniform
    assert e < 5e-2 * top, (e, top)                           # within 5 % of the largest logit of the fp64 oracle
    assert eq < 1.5 * eo + 1e-3, (eq, eo)                    # no further from the bf16-operand oracle than that oracle is from fp64
    assert abs(loss - float(rq["loss"]) * B) < 2e-3 * abs(loss)
    m.shutdown()


def _two_batches(O, batch_a):
    img_b, tgt_b, tge_b, nnz_b = O.synth_batch(5, 36, seed=4321, max_len=6, min_len=2)
    return batch_a, [img_b, tgt_b, tge_b, nnz_b, batch_a[4]]


@pytest.mark.parametrize("sharp,lr,bound", [(False, 0.02, 2e-3), (False, 0.1, None)])
def test_ten_step_trajectory_fp32(cuda, sharp, lr, bound):
    """Ten FREE-RUNNING optimisation steps (model.lua:695-706: feval, then optim.sgd_list with the per-group clip at 5) on two alternating
    batches, HIP (Model.step) against the oracle (train_step_manual + sgd_list), B = 5: per-step loss, and after the tenth step every
    parameter tensor and the BatchNorm running statistics.
    The training dynamics themselves amplify a perturbation -- measured in round 4 at the reference's lr = 0.1: the loss difference between
    the fp32 path and the fp64 oracle sits at fp32 resolution (<= 7e-8 relative) for four steps, jumps to 1e-5 when a ReLU / arg-max
    near-tie falls differently, and then grows ~4x per step (BatchNorm over 5 x 8 positions) -- a property of the optimisation problem, not
    of either implementation.  So the free runs are followed, REPORTED and held to bounds that a near-tie cannot break (lr = 0.02: every
    parameter within 2e-3 and every loss within 1e-3 relative; lr = 0.1: 2e-2), and the per-step agreement at lr = 0.1 along a real
    trajectory is what test_trajectory_step_by_step_fp32 holds STRICTLY (2e-5 per step).  (Measured at lr = 0.02 on two boxes: worst
    parameter error 3.9e-7 after ten steps when no near-tie fell differently, 2.2e-4 when one did at step 6 -- the split-K atomics make the
    last bits of a filter gradient run-dependent, so which of the two happens is not reproducible; lr = 0.1 -> 2.9e-4.  The SHARPENED weights are not run freely at all: their gradient norms are 200-750
    per group, every step is clipped, and a 1e-6 parameter difference is a 5e-4 loss difference one step later -- measured 28 % loss
    difference after ten free steps at lr = 0.02 while every single step from the oracle's state agrees to 6e-5.)"""
    m, O, ocfg, P0, st0, batch_a = make(SMALL, B=5, W=36, maxlen=6)
    Pinit = O.sharpen_params(P0) if sharp else P0
    P = Pinit
    m.set_parameters(P, st0)
    m.optim_state = {"learningRate": lr}
    batch_a, batch_b = _two_batches(O, batch_a)
    st = {k: v.clone() for k, v in st0.items()}
    clipped, worst_loss = 0, 0.0
    for step in range(10):
        batch = batch_a if step % 2 == 0 else batch_b
        img, tgt, tge = tensors(batch)
        loss_ref, G, _, st = O.train_step_manual(P, st, ocfg, img, tgt, tge)
        P, norms = O.sgd_list(P, G, lr, 5.0)
        clipped += sum(1 for n in norms if n[1] > 5.0)
        loss, _ = m.step(batch, False)
        rel = abs(loss - float(loss_ref) * 5) / abs(loss)
        worst_loss = max(worst_loss, rel)
        print(f"[trajectory] sharp={sharp} lr={lr} step {step}: loss hip {loss:.6f} oracle {float(loss_ref) * 5:.6f} (rel {rel:.1e}); gradient norms per group {[round(float(n[1]), 3) for n in norms]}")
    got = m.get_parameters()
    worst = ("", 0.0)
    for k, v in P.items():
        e = (got[k].double() - v).abs().max().item()
        if e > worst[1]: worst = (k, e)
    moved = max((P[k] - Pinit[k]).abs().max().item() for k in P)
    bn = m.get_bn_state()
    ebn = max((bn[k].double() - v).abs().max().item() for k, v in st.items())
    print(f"[trajectory] sharp={sharp} lr={lr}: after 10 steps the parameters moved by up to {moved:.3f}; worst parameter error {worst[1]:.3e} ({worst[0]}), "
          f"worst per-step loss difference {worst_loss:.1e} relative, running statistics {ebn:.1e}; {clipped} group clips were active")
    assert moved > 1e-3
    if bound is not None:
        assert worst[1] < bound and worst_loss < 1e-3 and ebn < 1e-2, (worst, worst_loss, ebn)
    else:
        assert worst[1] < 2e-2 and worst_loss < 2e-2, (worst, worst_loss)      # followed and reported (docstring); a broken update is off by O(1)
    m.shutdown()


@pytest.mark.parametrize("sharp", [False, True])
def test_trajectory_step_by_step_fp32(cuda, sharp):
    """The same ten steps at the reference's lr = 0.1, held STRICTLY step by step: before step k the HIP model is given the oracle's
    parameters and running statistics of step k (so the comparison happens at ten different points of a real trajectory, with the clip
    active under sharpening), takes ONE step through Model.step, and must land on the oracle's step k + 1: every parameter within 2e-5
    (sharpened: 2e-4 -- the clipped step has length lr x 5 = 0.5 per group whatever the gradient's size, so a 1e-4 relative error of the
    gradient DIRECTION is a 5e-5 absolute error of the parameters; measured 6.3e-5), running statistics within 1e-5, loss within 1e-5
    relative."""
    m, O, ocfg, P0, st0, batch_a = make(SMALL, B=5, W=36, maxlen=6)
    P = O.sharpen_params(P0) if sharp else P0
    batch_a, batch_b = _two_batches(O, batch_a)
    st = {k: v.clone() for k, v in st0.items()}
    worst = ("", 0.0, -1)
    for step in range(10):
        batch = batch_a if step % 2 == 0 else batch_b
        img, tgt, tge = tensors(batch)
        m.set_parameters(P, st)
        loss_ref, G, _, st = O.train_step_manual(P, st, ocfg, img, tgt, tge)
        P, norms = O.sgd_list(P, G, 0.1, 5.0)
        loss, _ = m.step(batch, False)
        assert abs(loss - float(loss_ref) * 5) < 1e-5 * abs(loss), (step, loss, float(loss_ref) * 5)
        got = m.get_parameters()
        for k, v in P.items():
            e = (got[k].double() - v).abs().max().item()
            if e > worst[1]: worst = (k, e, step)
            assert e < (2e-4 if sharp else 2e-5), (step, k, e)
        bn = m.get_bn_state()
        for k, v in st.items():
            assert (bn[k].double() - v).abs().max().item() < 1e-5, (step, k)
    print(f"[trajectory] sharp={sharp}: ten single steps along the oracle's lr = 0.1 trajectory, worst parameter error {worst[1]:.3e} ({worst[0]}, step {worst[2]})")
    m.shutdown()
