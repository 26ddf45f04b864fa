"""CPU: host-side logic of the Model mirror (utils.lua string/metric helpers, synthetic generator, bench FLOP model)
and the data-parallel exchange on the gloo backend with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generators_agree():
    import aocr
    import oracle_torch as O
    assert np.array_equal(O.counter_uniform(910820, 3, 1000), aocr.synth.counter_uniform(910820, 3, 1000))
    assert np.array_equal(O.counter_normal(1, 2, 77), aocr.synth.counter_normal(1, 2, 77))
    a, b = O.synth_batch(5, 64, max_len=7), aocr.synth.synth_batch(5, 64, max_len=7)
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    u = aocr.synth.counter_uniform(0, 0, 4)
    assert u.min() >= 0 and u.max() < 1


def test_word_error_rate_and_vocab():
    import aocr
    # utils.lua:104-134: 0-9 -> ids 4..13, a-z -> 14..39
    assert aocr.numlist2str([4, 13, 14, 39]) == "09az"
    labels = np.array([[14, 15, 3, 1, 1], [14, 15, 16, 3, 1], [3, 1, 1, 1, 1], [20, 20, 20, 20, 20]])
    gold = np.array([[14, 15, 3, 1, 1], [14, 15, 3, 1, 1], [3, 9, 9, 9, 9], [20, 20, 20, 20, 3]])
    werr, pred, gl = aocr.eval_word_err_rate(labels, gold, True)
    assert werr == 2 and pred[:2] == ["ab", "abc"] and gl[3] == "gggg"      # cut at first EOS, exact match (utils.lua:136-175)


def test_bench_flop_model_matches_survey():
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    assert abs(b.flops_per_image(100, 256, 1, 2, 24) / 1e9 - 1.5524) < 1e-3      # SURVEY.md 8(d) C2 forward
    assert abs(b.flops_per_image(256, 256, 1, 2, 24) / 1e9 - 3.6048) < 1e-3      # C3 forward


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "torch-attention-ocr_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    torch.set_num_threads(2)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aocr.dist as adist
    import oracle_torch as O
    cfg = O.OcrConfig(enc_hidden=8, dec_layers=2, input_feed=True)
    P, st = O.init_params(cfg, 5), O.init_bn_state()
    img, t, te, _ = O.synth_batch(4, 36, max_len=4, min_len=2)
    img, t, te = torch.from_numpy(img), torch.from_numpy(t), torch.from_numpy(te)
    sl = slice(rank * 2, rank * 2 + 2)                                   # contiguous slice of the global batch

    def grads(im, tt, tte, scale):
        Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        r = O.forward_train(Pg, {k: v.clone() for k, v in st.items()}, cfg, im, tt, tte, training=False, grad_through_quirk=True)
        nll_sum = r["loss"] * im.shape[0]
        (nll_sum * scale).backward()
        flat = torch.cat([Pg[k].grad.reshape(-1) if Pg[k].grad is not None else torch.zeros(Pg[k].numel(), dtype=torch.float64)
                          for k in Pg])
        return flat, nll_sum.detach().reshape(1)
    g, loss = grads(img[sl], t[sl], te[sl], adist.grad_scale(2))         # 1/(local * world) = 1/global
    adist.exchange(g, loss)                                              # the one collective of the hot path
    gfull, lfull = grads(img, t, te, 1.0 / 4)
    q.put((rank, float((g - gfull).abs().max()), float((loss - lfull).abs().max()), adist.world_size()))
    dist.destroy_process_group()


def test_data_parallel_exchange_gloo_world2():
    """DP-2 == single process on the concatenated batch (BatchNorm in eval mode so samples are independent)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
    for rank, gerr, lerr, w in res:
        assert w == 2 and gerr < 1e-12 and lerr < 1e-10, (rank, gerr, lerr)
