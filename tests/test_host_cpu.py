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
    assert abs(b.flops(100, 256, 1, 2, 24)["total"] / 1e9 - 1.5524) < 1e-3      # SURVEY.md 8(d) C2 forward
    assert abs(b.flops(256, 256, 1, 2, 24)["total"] / 1e9 - 3.6048) < 1e-3      # C3 forward


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
    g2, loss2 = g.clone(), loss.clone()
    adist.exchange(g, loss)                                              # the one collective of the hot path
    # the bucketed form used on the GPU (buckets back to front, in the order the backward pass completes them)
    n = g2.numel(); cut = [0, n // 7, n // 3, (3 * n) // 4, n]
    ranges = [(cut[3], cut[4]), (cut[2], cut[3]), (cut[1], cut[2]), (cut[0], cut[1])]
    waited = []
    adist.exchange_overlapped(g2, loss2, ranges, lambda k, stream: waited.append(k), None)
    assert waited == [0, 1, 2, 3] and torch.equal(g2, g) and torch.equal(loss2, loss)
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


def test_gradient_buckets_partition_the_flat_vector():
    """aocr_grad_buckets (host-only): four disjoint ranges covering the flat vector, listed in the order the backward pass
    completes them -- decoder + projector, encoders, CNN from conv5 upwards, conv1..conv4."""
    import aocr
    from aocr import dist as adist
    from aocr._lib import Config
    cfg = Config(batch_size=4, img_h=32, max_img_w=100, enc_hidden=256, enc_layers=1, dec_layers=2, vocab=39, emb=20, input_feed=1,
                 max_decoder_l=50, max_beam=1, compute=1)
    table, counts = aocr.param_table(cfg)
    off = {name: o for name, g, o, shape in table}
    n = sum(counts)
    r = adist.bucket_ranges(cfg)
    assert r[0] == (off["dec.lookup"], n) and r[1] == (off["enc_fw.l1.i2h.w"], off["dec.lookup"])
    assert r[2] == (off["cnn.conv5.w"], off["enc_fw.l1.i2h.w"]) and r[3] == (0, off["cnn.conv5.w"])
    covered = sorted(r)
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
