"""Observability rows of SURVEY.md section 8 (f-4): the `results.txt` lines of a `-phase test` step (model.lua:628-633) and the per-group
norm lines of optim.sgd_list (optim_sgd.lua:49), both compared with what the oracle's restatement of those lines prints."""
import numpy as np
import pytest
import torch

from test_step_gpu import CASES, make

pytestmark = pytest.mark.gpu


def _oracle_results_lines(O, ref, tge, paths, max_decoder_l):
    """model.lua:628-633 on the oracle's decode: '%s\\t%s\\t%s\\t%f\\t%f\\n' of (path, gold string, predicted string, score, gold score); the strings are
    evalWordErrRate's (utils.lua:136-175): both label rows cut at the first EOS, numlist2str of the rest (PAD = id 1 maps to '.', chr(46), like the reference)."""
    import dict_oracle as D
    B = len(paths)
    tge_pad = np.full((B, max_decoder_l), 1, dtype=np.int64)
    tge_pad[:, :tge.shape[1]] = np.asarray(tge)
    _, pred, gold, _, _ = D.eval_word_err_rate(ref["labels"].numpy(), tge_pad)
    return ["%s\t%s\t%s\t%f\t%f\n" % (paths[i], gold[i], pred[i], float(ref["scores"][i]), float(ref["gold_scores"][i])) for i in range(B)]


@pytest.mark.parametrize("case,beam", [(0, 1), (3, 5)])
def test_results_txt_lines_match_oracle(cuda, tmp_path, case, beam):
    m, O, ocfg, P, st, batch = make(CASES[case], B=4, W=36, maxlen=5, max_decoder_l=10)
    st = {k: (v + 0.05 if k.endswith("rm") else v * 1.3) for k, v in st.items()}
    m.set_parameters(P, st)
    m.vis(str(tmp_path))                                                   # model:vis (model.lua:708-718) opens <dir>/results.txt
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    ref = O.decode_beam(P, st, ocfg, img, tgt, tge, beam=beam, max_decoder_l=10)
    for _ in range(2):                                                     # two steps: the file is appended to and flushed per step
        m.step(batch, True, beam)
    m.shutdown()                                                           # closes the file (model.lua:727-731)
    got = open(tmp_path / "results.txt").read().splitlines(keepends=True)
    want = _oracle_results_lines(O, ref, batch[2], batch[4], 10)
    assert len(got) == 2 * len(want)
    for half in (got[:4], got[4:]):
        for g, w in zip(half, want):
            gp, wp = g.rstrip("\n").split("\t"), w.rstrip("\n").split("\t")
            assert len(gp) == 5 and gp[:3] == wp[:3], (g, w)               # path, gold string, predicted string: exact
            assert abs(float(gp[3]) - float(wp[3])) < 2e-3 and abs(float(gp[4]) - float(wp[4])) < 2e-3, (g, w)
            assert all(len(x.split(".")[1]) == 6 for x in gp[3:])          # '%f': six decimals
    print(f"[parity] results.txt case {case} beam {beam}: {got[0]!r}")


def test_vis_reports_unwritable_directory(cuda, tmp_path, capsys):
    """model.lua:711-717: a results file that cannot be created prints the error and switches visualisation off instead of raising."""
    m, O, ocfg, P, st, batch = make(CASES[0], B=4, W=36, maxlen=5, max_decoder_l=10)
    m.vis(str(tmp_path / "missing" / "dir"))
    assert "cannot be created" in capsys.readouterr().out and m.visualize is False and m.visualize_file is None
    m.step(batch, True, 1)                                                 # decodes without writing anything
    m.shutdown()


@pytest.mark.parametrize("scale", [1.0, 40.0])                             # 40 x: every group's gradient norm above the clip threshold of 5
def test_format_norms_matches_oracle_group_norms(cuda, scale):
    m, O, ocfg, P, st, batch = make(CASES[0], B=4, W=36, maxlen=5)
    assert m.format_norms() == []                                          # nothing to report before the first step
    if scale != 1.0:
        P = {k: (v * scale if k.startswith("proj.") else v) for k, v in P.items()}
        m.set_parameters(P, st)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    _, G, _, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    _, norms = O.sgd_list(P, G, 0.1)
    m.step(batch, False)
    lines = m.format_norms()
    assert len(lines) == 5
    for i, (line, (pn, gn)) in enumerate(zip(lines, norms)):
        head, p_s, g_s = line.split(", ")
        assert head == "i: %d" % (i + 1) and p_s.startswith("param norm: ") and g_s.startswith("grad norm: "), line
        gp, gg = float(p_s.split(": ")[1]), float(g_s.split(": ")[1])
        assert abs(gp - pn) <= 2e-5 * max(1.0, pn) and abs(gg - gn) <= 2e-4 * max(1.0, gn), (line, pn, gn)    # '%f' of an fp32 norm against the fp64 one
    print("[parity] norms: " + " | ".join(lines))
    m.shutdown()
