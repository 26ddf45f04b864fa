"""Dictionary-constrained decoding and word scoring on the GPU, through the C ABI, against
oracle/dict_oracle.py: the selection kernel alone on random scores and random tries, the edit-distance kernel, and the whole
forward_only step with -use_dictionary against the fp64 oracle's decode."""
import ctypes as C
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

V = 39


def _words(rng, n, alphabet, lo, hi):
    return ["".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(n)]


def _node_map(trie, root):
    """flat node id -> oracle table (by simultaneous walk)."""
    seen = {0: root}; stack = [0]
    while stack:
        n = stack.pop()
        for v in range(1, 65):
            c = trie.next(n, v)
            if c is not None and c not in seen:
                seen[c] = seen[n][v]; stack.append(c)
    return seen


@pytest.mark.parametrize("kin,kout,nwords,digit_prefix", [(1, 1, 40, False), (1, 5, 40, False), (1, 5, 2, False), (5, 5, 40, False),
                                                          (3, 3, 6, True), (5, 5, 400, True), (2, 2, 1, False)])
def test_beam_select_dict_matches_oracle(cuda, kin, kout, nwords, digit_prefix):
    import aocr
    import dict_oracle as DO
    from aocr import check, lib, ptr
    rng = random.Random(kin * 100 + kout * 10 + nwords)
    words = _words(rng, nwords, "abcdefghij0123", 1, 4)
    trie = aocr.build_trie(words, digit_prefix).to(cuda)
    root = DO.load_dictionary(words, digit_prefix)
    tables = _node_map(trie, root)
    ids = {id(d): n for n, d in tables.items()}
    B = 37
    g = torch.Generator().manual_seed(5)
    logp = torch.log_softmax(torch.randn(B * kin, V, generator=g) * 2, dim=1)
    logp[0, :] = logp[0, 0]                                        # a row of exact ties: lowest index first
    first = kin == 1
    scores_in = torch.randn(B, kin, generator=g)
    prev = torch.randint(1, V + 1, (B * kin,), generator=g).int()
    nodes_in = [[rng.randrange(trie.n_nodes) for _ in range(kin)] for _ in range(B)]
    # oracle
    exp_tok, exp_par, exp_sc, exp_loc = [], [], [], []
    for b in range(B):
        if first:
            toks, sc, new = DO.select_first(logp[b].tolist(), root, kout)
            pars = [0] * kout
        else:
            lp = logp[b * kin:(b + 1) * kin].clone()
            for j in range(kin):
                if int(prev[b * kin + j]) in (1, 3):
                    lp[j, 0] = 0.0
            total = (lp + scores_in[b].unsqueeze(1)).reshape(-1).tolist()
            toks, raws, sc, new = DO.select_next(total, [tables[n] for n in nodes_in[b]], kout, V)
            pars = [r // V for r in raws]
        exp_tok.append(toks); exp_par.append(pars); exp_sc.append(sc); exp_loc.append([ids[id(d)] for d in new])
    # device
    d_logp = logp.to(cuda); d_scores = torch.zeros(B, max(kin, kout), device=cuda)
    d_sc = (scores_in if not first else torch.zeros(B, kout)).contiguous().to(cuda)
    if not first and kout != kin:
        pytest.skip("kin == kout after the first step")
    d_tok = torch.zeros(B, kout, dtype=torch.int32, device=cuda); d_par = torch.zeros_like(d_tok); d_out = torch.zeros_like(d_tok)
    d_in = torch.tensor(nodes_in, dtype=torch.int32, device=cuda)
    desc = trie.desc()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.aocr_beam_select_dict(s, ptr(d_logp), None if first else ptr(prev.to(cuda)), ptr(d_sc), ptr(d_tok), ptr(d_par), B, kin, kout, V,
                                    C.byref(desc), None if first else ptr(d_in), ptr(d_out)), "aocr_beam_select_dict")
    torch.cuda.synchronize()
    assert d_tok.cpu().tolist() == exp_tok
    assert d_par.cpu().tolist() == exp_par
    assert d_out.cpu().tolist() == exp_loc
    assert torch.allclose(d_sc.cpu(), torch.tensor(exp_sc, dtype=torch.float32), atol=1e-6)


def test_beam_select_dict_rejects_bad_arguments(cuda):
    import aocr
    from aocr import lib, ptr
    trie = aocr.build_trie(["ab"]).to(cuda)
    desc = trie.desc()
    x = torch.zeros(4, V, device=cuda); sc = torch.zeros(4, device=cuda); ti = torch.zeros(4, dtype=torch.int32, device=cuda)
    assert lib.aocr_beam_select_dict(None, ptr(x), None, ptr(sc), ptr(ti), ptr(ti), 4, 1, 1, V, None, None, ptr(ti)) != 0
    assert lib.aocr_beam_select_dict(None, ptr(x), None, ptr(sc), ptr(ti), ptr(ti), 4, 1, 1, 65, C.byref(desc), None, ptr(ti)) != 0
    assert b"64" in lib.aocr_last_error()


@pytest.mark.parametrize("B,L", [(1, 1), (70, 10), (256, 50), (3, 200)])
def test_edit_distance_kernel(cuda, B, L):
    import dict_oracle as DO
    from aocr.dictionary import edit_distance_device
    rng = np.random.default_rng(B * 1000 + L)
    tgt = rng.integers(4, 40, size=(B, L)).astype(np.int32)
    lab = tgt.copy()
    for b in range(B):
        n = int(rng.integers(0, L + 1))
        if n < L:
            tgt[b, n] = 3
        kind = b % 5
        lab[b] = tgt[b]
        if kind == 1:                                   # substitutions
            idx = rng.integers(0, L, size=3); lab[b, idx] = rng.integers(4, 40, size=3)
        elif kind == 2:                                 # deletion: shift left
            lab[b, :-1] = tgt[b, 1:]; lab[b, -1] = 3
        elif kind == 3:                                 # unrelated
            lab[b] = rng.integers(3, 40, size=L)
        elif kind == 4:                                 # empty prediction
            lab[b, 0] = 3
    wo, po, go, dist, tlen = DO.eval_word_err_rate(lab, tgt)
    d, t = edit_distance_device(torch.from_numpy(lab).to(cuda), torch.from_numpy(tgt).to(cuda))
    assert d.cpu().tolist() == dist
    assert t.cpu().tolist() == tlen
    assert int((d != 0).sum()) == int(wo)


@pytest.mark.parametrize("case,beam,digit_prefix", [(0, 1, False), (0, 5, False), (1, 3, False), (2, 3, True), (3, 5, True)])
def test_decode_with_dictionary_small(cuda, case, beam, digit_prefix):
    import aocr
    import dict_oracle as DO
    from test_step_gpu import CASES, make
    m, O, ocfg, P, st, batch = make(CASES[case], B=4, W=36, maxlen=5, max_decoder_l=10)
    st = {k: (v + 0.05 if k.endswith("rm") else v * 1.3) for k, v in st.items()}
    P = dict(P); P["proj.b"] = P["proj.b"].clone(); P["proj.b"][:3] -= 4.0     # random weights: make PAD/GO/EOS unlikely, so words get spelled
    m.set_parameters(P, st)
    rng = random.Random(case * 10 + beam)
    words = _words(rng, 25, "abcdefghijklmnopqrstuvwxyz0123456789", 2, 6)
    trie = aocr.build_trie(words, digit_prefix)
    root = DO.load_dictionary(words, digit_prefix)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    ref = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=10, trie=root)
    free = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=10)
    loss, stats = m.step(batch, True, beam, trie)
    out = m._dec_out
    print(f"[parity] dict decode case {case} beam {beam}: labels[0] {out.labels[0].tolist()} (unconstrained {free['labels'][0].tolist()})")
    assert np.array_equal(out.labels, ref["labels"].numpy().astype(np.int32))
    if not digit_prefix:                               # (with the digit prefix a run of digits is admissible as it is)
        assert not np.array_equal(out.labels, free["labels"].numpy().astype(np.int32))     # the constraint did something
    assert np.abs(out.scores - ref["scores"].numpy()).max() < 2e-3
    assert np.abs(out.gold_scores - ref["gold_scores"].numpy()).max() < 2e-3
    assert stats[1] == ref["num_correct"]
    # every decoded row is a dictionary path: word, EOS, then PAD only (without the digit prefix)
    if not digit_prefix:
        for row in out.labels:
            ids = row.tolist()
            cut = ids[:ids.index(3) + 1] if 3 in ids else [v for v in ids if v != 1]
            assert trie.walk([v for v in cut if v != 1]) is not None, ids
    m.shutdown()


def test_dictionary_decode_full_size_property(cuda):
    """C3 geometry (batch 256, 32x256, bf16), beam 3, a 20 k-word lexicon: whatever the random weights prefer, every decoded row
    must spell a path of the trie (PAD keeps the node, model.lua:469,502-503) -- a size-independent property of the constraint."""
    import aocr
    from test_step_gpu import make
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=256, W=256, maxlen=23,
                                    compute="bf16", max_decoder_l=30, max_beam=3)
    P = dict(P); P["proj.b"] = P["proj.b"].clone(); P["proj.b"][:3] -= 4.0
    m.set_parameters(P, st)
    rng = random.Random(1)
    words = _words(rng, 20000, "abcdefghijklmnopqrstuvwxyz", 3, 9)
    trie = aocr.build_trie(words).to(cuda)
    m.step(batch, True, 3, trie)
    labels = m._dec_out.labels
    assert labels.shape == (256, 30)
    distinct = set()
    for row in labels:
        ids = [int(v) for v in row if v != 1]                      # PAD never moves the node
        assert trie.walk(ids) is not None, ids
        distinct.add(tuple(ids))
    free_labels = None
    m.step(batch, True, 3)
    free_labels = m._dec_out.labels
    off_path = sum(trie.walk([int(v) for v in row if v != 1]) is None for row in free_labels)
    print(f"[property] dictionary decode at C3 size: 256/256 rows on a trie path ({len(distinct)} distinct); unconstrained: {256 - off_path}/256")
    assert off_path > 128                                          # the same weights without the constraint leave the lexicon
    m.shutdown()


@pytest.mark.gpu
def test_greedy_dictionary_decode_cluster_kernel(cuda, monkeypatch):
    """Greedy (beam 1) dictionary decode through the decoder cluster kernel's DEC variant (trie test + node update inside the kernel's
    selection) against the launch chain with project_select_kernel: same labels, every row on a trie path."""
    import aocr
    from test_step_gpu import make
    rng = random.Random(7)
    words = _words(rng, 5000, "abcdefghijklmnopqrstuvwxyz", 3, 9)
    out = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_DEC_CLUSTER", "1")
        else:
            monkeypatch.delenv("AOCR_NO_DEC_CLUSTER", raising=False)
        m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=64, W=128, maxlen=8,
                                        compute="bf16", max_decoder_l=16, max_beam=1)
        P = dict(P); P["proj.w"] = P["proj.w"] * 40.0; P["proj.b"] = P["proj.b"].clone(); P["proj.b"][:3] -= 4.0
        m.set_parameters(P, st)
        trie = aocr.build_trie(words).to(cuda)
        m.step(batch, True, 1, trie)
        out[knob] = (np.array(m._dec_out.labels), np.array(m._dec_out.scores))
        for row in out[knob][0]:
            assert trie.walk([int(v) for v in row if v != 1]) is not None
        m.shutdown()
    same = (out["0"][0] == out["1"][0]).all(axis=1)
    print(f"[parity] greedy dictionary decode, cluster vs chain: identical rows {same.mean():.3f}")
    assert same.mean() >= 0.9
    assert np.abs(out["0"][1] - out["1"][1])[same].max() < 2e-2 * max(1.0, np.abs(out["1"][1]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("beam", [3, 5])
def test_beam_dictionary_decode_chain_kernel(cuda, monkeypatch, beam):
    """Beam search with -use_dictionary inside the decoder chain kernel's BEAM variant (trie test per candidate, node of every surviving hypothesis,
    the repeat-the-best rule when fewer candidates than beams are admissible: model.lua:405-445,460-513) against the launch chain with
    project_select_kernel: same labels, every row on a trie path."""
    import aocr
    from test_step_gpu import make
    rng = random.Random(11)
    words = _words(rng, 5000, "abcdefghijklmnopqrstuvwxyz", 3, 9)
    out = {}
    for knob in ("0", "1"):
        if knob == "1":
            monkeypatch.setenv("AOCR_NO_DEC_CHAINS_BEAM", "1")
        else:
            monkeypatch.delenv("AOCR_NO_DEC_CHAINS_BEAM", raising=False)
        m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=64, W=128, maxlen=8,
                                        compute="bf16", max_decoder_l=16, max_beam=beam)
        P = dict(P); P["proj.w"] = P["proj.w"] * 40.0; P["proj.b"] = P["proj.b"].clone(); P["proj.b"][:3] -= 4.0
        m.set_parameters(P, st)
        trie = aocr.build_trie(words).to(cuda)
        m.step(batch, True, beam, trie)
        assert int(m.get_tensor("cl_err").view(torch.int32)[0]) == 0
        out[knob] = (np.array(m._dec_out.labels), np.array(m._dec_out.scores))
        for row in out[knob][0]:
            assert trie.walk([int(v) for v in row if v != 1]) is not None
        m.shutdown()
    same = (out["0"][0] == out["1"][0]).all(axis=1)
    print(f"[parity] beam-{beam} dictionary decode, chain kernel vs launch chain: identical rows {same.mean():.3f}")
    assert same.mean() >= 0.9
    assert np.abs(out["0"][1] - out["1"][1])[same].max() < 2e-2 * max(1.0, np.abs(out["1"][1]).max())
