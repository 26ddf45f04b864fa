"""Checkpoints on the GPU path: Model.load of a Torch7-serialized reference checkpoint (restated object
tree, tests/t7_fixtures.py) must give the decoder logits the fp64 oracle computes from the same named weights; Model.save to
`.t7` and back must reproduce parameters, BatchNorm statistics, step and optimizer state exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Le,Ld,feed,names", [(1, 2, True, True), (2, 3, True, False), (1, 1, False, False)])
def test_load_reference_t7_then_logits_match_oracle(cuda, tmp_path, Le, Ld, feed, names):
    import aocr
    import oracle_torch as O
    from aocr import t7
    from t7_fixtures import random_params, reference_checkpoint
    He = 32
    P, S, config = random_params(He, Le, Ld, feed, seed=3)
    path = str(tmp_path / "model-4321")                               # the reference's files carry no extension (train.lua)
    t7.save(path, reference_checkpoint(P, S, config, global_step=4321, lr=0.0125, names=names), cuda=True)
    m = aocr.Model().load(path, dict(batch_size=4, max_img_w=36, max_decoder_l=12, max_beam=1))
    assert m.global_step == 4321 and m.optim_state == {"learningRate": 0.0125}
    assert m.encoder_num_layers == Le and m.decoder_num_layers == Ld and m.input_feed == bool(feed)
    ocfg = O.OcrConfig(enc_hidden=He, enc_layers=Le, dec_layers=Ld, input_feed=bool(feed))
    Pt = {k: torch.from_numpy(v).double() for k, v in P.items()}
    St = {k: torch.from_numpy(v).double() for k, v in S.items()}
    img, tgt, tge, nnz = O.synth_batch(4, 36, max_len=5, min_len=2)
    batch = [img, tgt, tge, nnz, ["a", "b", "c", "d"]]
    with torch.no_grad():
        r = O.forward_train(Pt, {k: v.clone() for k, v in St.items()}, ocfg, torch.from_numpy(np.asarray(img)), torch.from_numpy(np.asarray(tgt)),
                            torch.from_numpy(np.asarray(tge)), training=False)
    logits, loss = m.forward_logits(batch, training=False)
    e = (logits.double() - r["logits"]).abs().max().item()
    print(f"[parity] t7 checkpoint Le={Le} Ld={Ld} feed={feed}: logits max-abs {e:.3e}")
    assert e < 1e-4
    m.shutdown()


@pytest.mark.parametrize("layout", ["reference", "flat"])
def test_save_t7_round_trip(cuda, tmp_path, layout):
    """Model.save to Torch7 serialization in the reference's own layout (the five nets as nn / nngraph trees, model.lua:720-725) and in
    the flat layout; Model.load reads both back bit for bit."""
    import aocr
    from test_step_gpu import CASES, make
    m, O, ocfg, P, st, batch = make(CASES[0], B=4, W=36, maxlen=5)
    m.train_forward_backward(batch); m.sgd_step()
    m.global_step = 17; m.optim_state = {"learningRate": 0.05}
    path = str(tmp_path / "model.t7")
    m.save(path, layout=layout)
    from aocr import t7
    assert isinstance(t7.load(path).get(1), dict) == (layout == "reference")
    m2 = aocr.Model().load(path, dict(batch_size=4, max_img_w=36, max_decoder_l=12, max_beam=5))
    assert m2.global_step == 17 and m2.optim_state == {"learningRate": 0.05}
    assert torch.equal(m.params.cpu(), m2.params.cpu()) and torch.equal(m.bn_state.cpu(), m2.bn_state.cpu())
    l1, _ = m.forward_logits(batch); l2, _ = m2.forward_logits(batch)
    assert torch.equal(l1, l2)
    m.shutdown(); m2.shutdown()
