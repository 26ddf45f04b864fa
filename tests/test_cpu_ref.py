"""The C++/OpenMP restatement (oracle/cpu_ref) against the committed golden fixtures and the Python oracle: two independently written
restatements of the reference's Lua (one on PyTorch conv / batch_norm / autograd, one spelled out as im2col + GEMM, explicit
BatchNorm / pooling / BPTT loops) must agree to float64 rounding.  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle", "cpu_ref"))
import cpu_ref as R  # noqa: E402
import oracle_torch as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {
    "feed_ld2": (dict(enc_hidden=16, enc_layers=1, dec_layers=2, input_feed=True), 2, 36, 5),
    "nofeed_ld2": (dict(enc_hidden=16, enc_layers=1, dec_layers=2, input_feed=False), 2, 36, 5),
    "feed_ld1": (dict(enc_hidden=16, enc_layers=1, dec_layers=1, input_feed=True), 2, 36, 5),
    "le2_ld3": (dict(enc_hidden=16, enc_layers=2, dec_layers=3, input_feed=True), 2, 36, 5),
}


def setup(kw, B, W, maxlen, H=32, seed=910820):
    cfg = O.OcrConfig(**kw)
    P, st = O.init_params(cfg, seed), O.init_bn_state()
    names = [s[0] for s in O.param_spec(cfg)]
    spec = [(s[0], s[1]) for s in O.param_spec(cfg)]
    flat = R.flatten({k: v.numpy() for k, v in P.items()}, names)
    bn = np.concatenate([st[f"cnn.bn{i}.{k}"].numpy() for i in (3, 5, 7) for k in ("rm", "rv")])
    img, tgt, tge, nnz = O.synth_batch(B, W, max_len=maxlen, min_len=2, H=H)
    rc = R.make_cfg(cfg.enc_hidden, cfg.enc_layers, cfg.dec_layers, cfg.input_feed, img_h=H)
    return cfg, P, st, names, spec, flat, bn, img, tgt, tge, rc


def probes(t, n=32):
    f = np.asarray(t).reshape(-1)
    return f[(np.arange(n, dtype=np.int64) * 2654435761) % f.size]


@pytest.mark.parametrize("name", sorted(CASES))
def test_cpp_restatement_matches_golden(name):
    kw, B, W, maxlen = CASES[name]
    cfg, P, st, names, spec, flat, bn, img, tgt, tge, rc = setup(kw, B, W, maxlen)
    g = np.load(os.path.join(GOLD, name + ".npz"))
    r = R.train_step(rc, flat, bn, img, tgt, tge)
    assert abs(r["loss"] / B - float(g["loss"])) < 1e-10
    assert np.abs(r["feats"][:, :, :64] - g["feats_first"]).max() < 1e-10 and np.abs(r["feats"][:, :, -64:] - g["feats_last"]).max() < 1e-10
    assert np.abs(r["context"] - g["context"]).max() < 1e-10
    assert np.abs(r["logits"] - g["logits"]).max() < 1e-10
    G = R.unflatten(r["grads"], spec)
    for k in names:
        assert np.abs(probes(G[k]) - g["g:" + k]).max() < 1e-10 * max(1.0, np.abs(g["g:" + k]).max()), k
    o = 0
    for i, c in ((3, 256), (5, 512), (7, 512)):
        assert np.abs(r["bn_state"][o:o + c] - g[f"bn:cnn.bn{i}.rm"]).max() < 1e-12
        assert np.abs(r["bn_state"][o + c:o + 2 * c] - g[f"bn:cnn.bn{i}.rv"]).max() < 1e-12
        o += 2 * c
    # optim.sgd_list with clip 5 (inactive) and 0.05 (active)
    for clip, tag in ((5.0, "p5:"), (0.05, "p005:")):
        newp, norms = R.sgd(rc, flat, r["grads"], 0.1, clip)
        NP = R.unflatten(newp, spec)
        for k in names:
            assert np.abs(probes(NP[k]) - g[tag + k]).max() < 1e-10, (k, clip)
        assert np.abs(norms - g["norms"]).max() < 1e-9
    # decode with the updated parameters / running stats (greedy + beam 5), as gen_golden.py does
    newp, _ = R.sgd(rc, flat, r["grads"], 0.1, 5.0)
    for beam in (1, 5):
        d = R.decode(rc, newp, r["bn_state"], img, tgt, tge, beam, 8)
        assert np.array_equal(d["labels"], g[f"dec{beam}:labels"].astype(np.int32)), beam
        assert np.abs(d["scores"] - g[f"dec{beam}:scores"]).max() < 1e-9 and np.abs(d["gold_scores"] - g[f"dec{beam}:gold"]).max() < 1e-9
        assert abs(d["loss"] - float(g[f"dec{beam}:loss"])) < 1e-9


def test_cpp_restatement_matches_python_oracle_everywhere():
    """ALL entries of all gradient tensors (the fixtures hold 32 probes each), a ragged width, batch 3, tall strips (H = 64: View(512,-1)
    strings 3 rows of positions together)."""
    for kw, B, W, maxlen, H in ((dict(enc_hidden=24, enc_layers=2, dec_layers=2, input_feed=True), 3, 70, 6, 32),
                                (dict(enc_hidden=16, enc_layers=1, dec_layers=2, input_feed=True), 2, 40, 4, 64)):
        cfg, P, st, names, spec, flat, bn, img, tgt, tge, rc = setup(kw, B, W, maxlen, H=H)
        ti, tt, te = torch.from_numpy(img), torch.from_numpy(tgt), torch.from_numpy(tge)
        loss, G, aux, st2 = O.train_step_manual(P, st, cfg, ti, tt, te)
        r = R.train_step(rc, flat, bn, img, tgt, tge)
        assert abs(r["loss"] - float(loss) * B) < 1e-9
        assert np.abs(r["logits"] - aux["logits"].numpy()).max() < 1e-10
        Gc = R.unflatten(r["grads"], spec)
        for k in names:
            ref = G[k].numpy()
            assert np.abs(Gc[k] - ref).max() < 1e-10 * max(1.0, np.abs(ref).max()), k


def test_cpp_restatement_fp32_close():
    kw, B, W, maxlen = CASES["feed_ld2"]
    cfg, P, st, names, spec, flat, bn, img, tgt, tge, rc = setup(kw, B, W, maxlen)
    r64 = R.train_step(rc, flat, bn, img, tgt, tge)
    r32 = R.train_step(rc, flat, bn, img, tgt, tge, dtype=np.float32)
    assert np.abs(r32["logits"] - r64["logits"]).max() < 1e-4
    assert abs(r32["loss"] - r64["loss"]) < 1e-3
