"""CPU: the float64 oracle against its committed golden fixtures (tests/golden, made by oracle/gen_golden.py),
its two independent gradient derivations against each other, and hand-checkable leaf-op answers.
The reference has no tests or vectors of its own: parity is unpinned beyond this."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _gen():
    import importlib.util
    p = os.path.join(os.path.dirname(GOLD), "..", "oracle", "gen_golden.py")
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.abspath(p))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["feed_ld2", "nofeed_ld2", "feed_ld1", "le2_ld3", "sharp_feed_ld2", "sharp_le2_ld3"])
def test_oracle_matches_golden(name):
    g = _gen()
    kw, B, W, ml = g.CASES[name]
    got = g.run_case(kw, B, W, ml, name)
    ref = np.load(os.path.join(GOLD, f"{name}.npz"))
    assert set(got) == set(ref.files)
    for k in ref.files:
        a, b = np.asarray(got[k]), ref[k]
        if a.dtype.kind in "iu":
            assert np.array_equal(a, b), k
        else:
            assert np.allclose(a, b, rtol=1e-9, atol=1e-11), (k, np.abs(a - b).max())


def test_leaf_known_answers():
    ref = np.load(os.path.join(GOLD, "leaf_ops.npz"))
    # SpatialMaxPooling(kW=1,kH=2,dW=1,dH=2) halves the HEIGHT only (cnn.lua:29)
    assert ref["pool_kh2_kw1"].shape == (1, 1, 2, 4)
    assert np.array_equal(ref["pool_kh2_kw1"][0, 0], np.array([[4, 5, 6, 7], [12, 13, 14, 15]], dtype=np.float64))
    assert int(ref["conv7_width"]) == 24                       # 2x2 / pad 0 over width 25 -> T = W/4 - 1 (S7)
    lp = np.log(np.exp(0.2) / (np.exp(0.5) + np.exp(0.1) + np.exp(0.2)))
    assert abs(float(ref["nll_pad_weight0"]) - (-lp)) < 1e-12   # row 1 (PAD target) contributes 0
    assert np.allclose(ref["clip_13_to_5"], -np.array([3.0, 4.0, 12.0]) * 5.0 / 13.0)


@pytest.mark.parametrize("feed,Ld,Le", [(True, 2, 1), (False, 2, 1), (True, 1, 1), (True, 3, 2)])
def test_manual_bptt_equals_autograd(feed, Ld, Le):
    """model.lua:634-694 restated by hand vs PyTorch autograd over the same forward (incl. quirk S5)."""
    import oracle_torch as O
    cfg = O.OcrConfig(enc_hidden=8, enc_layers=Le, dec_layers=Ld, input_feed=feed)
    P, st = O.init_params(cfg, 7), O.init_bn_state()
    img, t, te, _ = O.synth_batch(3, 36, max_len=5, min_len=2)
    img, t, te = torch.from_numpy(img), torch.from_numpy(t), torch.from_numpy(te)
    la, Ga, _, _ = O.train_step_autograd(P, st, cfg, img, t, te)
    lm, Gm, _, _ = O.train_step_manual(P, st, cfg, img, t, te)
    assert abs(float(la) - float(lm)) < 1e-12
    for k in Ga:
        assert float((Ga[k] - Gm[k]).abs().max()) < 1e-12, k


def test_quirk_s5_initial_state():
    """-input_feed with >=2 decoder layers: h1(0) is zeroed, c1(0) = [c_fw(T); c_bw(1)] (model.lua:542-552)."""
    import oracle_torch as O
    cfg = O.OcrConfig(enc_hidden=8, dec_layers=2, input_feed=True)
    tr = {"enc_fw": (None, torch.ones(2, 8), torch.full((2, 8), 2.0)), "enc_bw": (None, torch.full((2, 8), 3.0), torch.full((2, 8), 4.0))}
    c, h = O.decoder_init_state(cfg, tr, 2, torch.zeros(1))
    assert torch.equal(c[0], torch.cat([torch.ones(2, 8), torch.full((2, 8), 3.0)], 1)) and float(h[0].abs().max()) == 0
    cfg1 = O.OcrConfig(enc_hidden=8, dec_layers=1, input_feed=True)
    c, h = O.decoder_init_state(cfg1, tr, 2, torch.zeros(1))
    assert torch.equal(h[0], torch.cat([torch.full((2, 8), 2.0), torch.full((2, 8), 4.0)], 1))


def test_synth_batch_layout():
    """data_gen.lua:107-117: targets = [GO ids PAD..], targets_eval = [ids EOS PAD..], num_nonzeros = sum(len+1)."""
    import oracle_torch as O
    img, t, te, nnz = O.synth_batch(6, 40, max_len=9)
    assert img.shape == (6, 1, 32, 40) and img.min() >= 0 and img.max() <= 255 and np.all(img == np.floor(img))
    assert t.shape == te.shape == (6, 10) and np.all(t[:, 0] == 2)
    for b in range(6):
        n = int((te[b] == 3).argmax())
        assert np.array_equal(t[b, 1:n + 1], te[b, :n]) and np.all(te[b, n + 1:] == 1) and np.all(t[b, n + 1:] == 1)
    assert nnz == int((te != 1).sum())


def test_s9_fixture_first_token_39():
    """Quirk S9 (model.lua:402-404,516): the oracle regenerates the committed fixture, whose two halves record the build's choice
    (parent = beam 1 at t = 1) and the reference's literal arithmetic.  Beam 5: identical (the replicas are identical at t = 1 and the
    recorded parent is never read).  Beam 1: the literal arithmetic gathers row b + 1 for the rows whose first token is id 39."""
    g = _gen()
    got = g.s9_fixture()
    ref = np.load(os.path.join(GOLD, "s9_first39.npz"))
    assert set(got) == set(ref.files)
    for k in ref.files:
        a, b = np.asarray(got[k]), ref[k]
        assert np.array_equal(a, b) if a.dtype.kind in "iu" else np.allclose(a, b, rtol=1e-9, atol=1e-11), k
    first = ref["b1:first_token"]
    rows39 = np.nonzero(first == 39)[0]
    assert 2 <= len(rows39) <= 4 and first[-1] != 39
    assert np.array_equal(ref["b1:fixed:src"], np.arange(6))
    want = np.arange(6); want[rows39] += 1
    assert np.array_equal(ref["b1:ref:src"], want)                                    # the next image's decoder state
    assert np.array_equal(ref["b5:fixed:labels"], ref["b5:ref:labels"]) and np.array_equal(ref["b5:fixed:scores"], ref["b5:ref:scores"])
    assert not np.allclose(ref["b1:fixed:scores"][rows39], ref["b1:ref:scores"][rows39])
    other = np.setdiff1d(np.arange(6), rows39)
    assert np.allclose(ref["b1:fixed:scores"][other], ref["b1:ref:scores"][other])


def test_s9_reference_mode_raises_for_last_row():
    """Torch7 raises an index error when the LAST row of a beam-1 batch emits id 39 first (row B + 1 does not exist)."""
    import oracle_torch as O
    g = _gen()
    cfg, P, st, img, tgt, tge = g.s9_inputs()
    P = dict(P); P["proj.b"] = P["proj.b"].clone(); P["proj.b"][38] += 1.0           # every row now emits id 39 first
    with pytest.raises(IndexError):
        O.decode_beam(P, st, cfg, img, tgt, tge, beam=1, max_decoder_l=8, s9="reference")
    d = O.decode_beam(P, st, cfg, img, tgt, tge, beam=1, max_decoder_l=8)            # the build's choice decodes every row
    assert (d["hist_tok"][0][:, 0] == 39).all()
