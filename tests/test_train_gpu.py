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
    first = last = None
    for step in range(500):
        if adadelta:
            loss = m.train_forward_backward(batch); m.adadelta_step()
        else:
            loss, _ = m.step(batch, False)
        per_tok = loss / nnz
        if step == 0:
            first = per_tok
        last = per_tok
    print(f"[train] {compute}{' adadelta' if adadelta else ''}: loss/token {first:.3f} -> {last:.4f}")
    assert abs(first - math.log(39)) < 0.6          # fresh parameters: close to the uniform distribution over 39 classes
    assert last < 0.1 * first and math.isfinite(last)      # (split-K atomics make the trajectory run-to-run different: loose bounds)
    for beam in (1, 3):
        loss, (n, correct) = m.step(batch, True, beam)
        print(f"[train] {compute}: forward_only beam {beam}: {correct:.0f}/{B} words right, gold-pass loss/token {loss / nnz:.4f}")
        assert correct >= B - 1
        assert int((m._dec_out.edit_distance != 0).sum()) == B - correct
    m.shutdown()
