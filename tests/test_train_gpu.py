"""System test of the train step on the GPU: the whole loop (feval + clipped SGD through the C ABI, model.lua:226-706 +
optim_sgd.lua:38-95) has to LEARN -- a fixed batch of synthetic line images is memorised, the per-token loss falls from
ln(39) towards zero and the forward-only step then decodes the labels it was trained on (greedy and beam, word accuracy
through the device edit-distance kernel)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compute,adadelta", [("f32", False), ("bf16", False), ("bf16", True)])
def test_overfits_a_fixed_batch(cuda, compute, adadelta):
    import aocr
    import oracle_torch as O
    B, W, L = 16, 64, 5
    img, tgt, tge, nnz = O.synth_batch(B, W, max_len=L - 1, min_len=2, seed=7)
    m = aocr.Model().create(dict(encoder_num_hidden=64, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=B,
                                 max_img_w=W, max_decoder_l=8, max_beam=3, compute=compute, learning_rate=0.1, seed=3))
    batch = [img, tgt, tge, nnz, [f"img{i}" for i in range(B)]]
    # plain SGD at lr 0.1 on 16 memorised lines spikes now and then (and split-K atomics make every run's trajectory different), so
    # the test trains until the per-token loss has stayed low for 40 steps in a row rather than sampling one fixed step
    first, per_tok, calm, steps = None, None, 0, 0
    while steps < 3000 and calm < 40:                # 40 calm steps: the BatchNorm running statistics (momentum 0.1) have settled too
        if adadelta:
            loss = m.train_forward_backward(batch); m.adadelta_step()
        else:
            loss, _ = m.step(batch, False)
        per_tok = loss / nnz
        first = per_tok if first is None else first
        calm = calm + 1 if per_tok < 0.02 else 0
        steps += 1
    print(f"[train] {compute}{' adadelta' if adadelta else ''}: loss/token {first:.3f} -> {per_tok:.4f} after {steps} steps")
    assert abs(first - math.log(39)) < 0.6          # fresh parameters: close to the uniform distribution over 39 classes
    assert calm >= 40 and math.isfinite(per_tok), (steps, per_tok)
    for beam in (1, 3):
        loss, (n, correct) = m.step(batch, True, beam)
        print(f"[train] {compute}: forward_only beam {beam}: {correct:.0f}/{B} words right, gold-pass loss/token {loss / nnz:.4f}")
        assert correct >= B - 2                      # eval-mode BatchNorm (running statistics) against train-mode memorisation
        assert int((m._dec_out.edit_distance != 0).sum()) == B - correct
    m.shutdown()
