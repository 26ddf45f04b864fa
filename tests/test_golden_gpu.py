"""GPU: the HIP path (through the C ABI) directly against the committed golden fixtures tests/golden/*.npz.  The oracle is
used only to regenerate the fixtures' inputs and initial weights (they are not stored: counter-based generator, seed in
oracle/gen_golden.py); every expected value -- features, context, decoder logits, loss, 32 probe entries of each of the 41
gradient tensors, the clipped SGD updates at two clip levels, the BatchNorm running statistics and the greedy / beam-5
decode of the updated model -- comes from the .npz files."""
import importlib.util
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "..", "oracle", "gen_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _probe(t, n=32):
    f = t.reshape(-1)
    idx = (np.arange(n, dtype=np.int64) * 2654435761) % f.numel()
    return f[torch.from_numpy(idx)].double().numpy()


@pytest.mark.parametrize("name", ["feed_ld2", "nofeed_ld2", "feed_ld1", "le2_ld3", "sharp_feed_ld2", "sharp_le2_ld3"])
def test_hip_matches_golden_fixture(cuda, name):
    import aocr
    g = _gen(); O = g.O
    kw, B, W, ml = g.CASES[name]
    ref = np.load(os.path.join(GOLD, f"{name}.npz"))
    cfg = O.OcrConfig(**kw)
    P, st = g.params_for(name, cfg), O.init_bn_state()          # sharp_*: the seeded weights scaled by oracle_torch.sharpen_params
    img, tgt, tge, nnz = O.synth_batch(B, W, max_len=ml, min_len=2)
    assert nnz == int(ref["nnz"])
    m = aocr.Model()
    m._set_structure(dict(encoder_num_hidden=cfg.enc_hidden, encoder_num_layers=cfg.enc_layers, decoder_num_layers=cfg.dec_layers,
                          input_feed=cfg.input_feed))
    m._set_runtime(dict(batch_size=B, max_img_w=W, max_decoder_l=8, max_beam=5, compute="f32"))
    m.optim_state = {"learningRate": 0.1}
    m._build()
    m.set_parameters(P, st)
    batch = [img, tgt, tge, nnz, [f"img{i}" for i in range(B)]]
    loss = m.train_forward_backward(batch)
    assert abs(loss - float(ref["loss"]) * B) < 1e-4 * max(1.0, abs(loss))
    feats = m.get_tensor("feats").transpose(0, 1).double().numpy()                    # (T,B,512) -> (B,T,512)
    assert np.abs(feats[:, :, :64] - ref["feats_first"]).max() < 2e-4 and np.abs(feats[:, :, -64:] - ref["feats_last"]).max() < 2e-4
    assert np.abs(m.get_tensor("context").double().numpy() - ref["context"]).max() < 1e-4
    lg = m.get_tensor("logits")[:, :, :cfg.vocab].double().numpy()
    e = np.abs(lg - ref["logits"]).max(); print(f"[parity] golden {name}: logits max-abs {e:.2e}")
    assert e < 1e-4                                                                   # BASELINE.json north_star tolerance
    assert e < 1e-4 * np.abs(ref["logits"]).max()                                     # ... and relative to the largest logit (tests/tol.py)
    if "attn_entropy_mean" in ref.files:
        print(f"[parity] golden {name}: sharpened regime, max |logit| {float(ref['logit_absmax']):.2f}, mean attention entropy {float(ref['attn_entropy_mean']):.2f} nat")
        assert float(ref["logit_absmax"]) > 1.0 and float(ref["attn_entropy_mean"]) < 1.0
    grads = m.get_gradients()
    worst = 0.0
    for k in grads:
        want = ref["g:" + k]
        scale = float(ref_scale := max(float(np.abs(want).max()), 1e-30))
        e = float(np.abs(_probe(grads[k]) - want).max())
        if k in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"):          # conv bias in front of a BatchNorm: exact gradient 0, rounding noise only
            w = k[:-1] + "w"                                            # ... measured against the size of the same layer's filter gradient
            assert ref_scale < 1e-9 and float(grads[k].abs().max()) < 1e-5 * float(grads[w].abs().max()), (k, float(grads[k].abs().max()))
            continue
        e /= max(scale, float(grads[k].abs().max()))
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    print(f"[parity] golden {name}: worst probed-gradient error {worst:.2e} of the tensor's largest entry")
    bn = m.get_bn_state()
    for k in bn:
        assert np.abs(bn[k].double().numpy() - ref["bn:" + k]).max() < 1e-5, k
    for clip, tag in ((5.0, "p5:"), (0.05, "p005:")):                                 # optim_sgd.lua:38-95
        m.set_parameters(P, st)
        m.train_forward_backward(batch)
        norms = m.sgd_step(lr=0.1, clip=clip)
        if clip == 5.0:
            assert np.abs(np.asarray(norms)[:, 0] - ref["norms"][:, 0]).max() < 1e-4 * max(1.0, ref["norms"][:, 0].max())
        got = m.get_parameters()
        for k in got:
            assert np.abs(_probe(got[k]) - ref[tag + k]).max() < 2e-5, (tag, k)
    # decode of the model after the clip-5 update with the updated running statistics (state left by the loop above is clip 0.05:
    # redo the clip-5 step)
    m.set_parameters(P, st)
    m.train_forward_backward(batch)
    m.sgd_step(lr=0.1, clip=5.0)
    for beam in (1, 5):
        loss_d, stats = m.step(batch, True, beam)
        out = m._dec_out
        assert np.array_equal(out.labels, ref[f"dec{beam}:labels"].astype(np.int32)), beam
        assert np.abs(out.scores - ref[f"dec{beam}:scores"]).max() < 2e-3 and np.abs(out.gold_scores - ref[f"dec{beam}:gold"]).max() < 2e-3
        assert abs(loss_d - float(ref[f"dec{beam}:loss"])) < 2e-3 * max(1.0, float(ref[f"dec{beam}:loss"]))
        assert stats[1] == int(ref[f"dec{beam}:correct"])
    print(f"[parity] golden {name}: SGD updates, BatchNorm statistics and greedy / beam-5 decode match the fixture")
    m.shutdown()
