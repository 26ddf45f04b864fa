"""Data path, GPU side: aocr_preprocess_lines (255*rgb2y + image.scale to 32 x W) against the numpy restatement, bit for bit
(every float op of the kernel is an explicitly rounded single-precision op in the oracle's order), over enlarging /
shrinking / equal sizes in both directions, gray and RGB sources; and aocr.data.DataGen end to end against the oracle's
DataGen on the same synthetic list."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))


@pytest.mark.parametrize("out_w", [100, 32, 256])
def test_preprocess_matches_oracle_bitwise(cuda, out_w):
    import aocr
    import data_oracle as D
    rng = np.random.default_rng(out_w)
    shapes = [(32, out_w, 1), (20, 37, 3), (64, 300, 3), (7, 500, 1), (48, out_w, 3), (32, 9, 1), (1, 1, 3), (33, 129, 1)]
    imgs = []
    for h, w, c in shapes:
        a = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        imgs.append(a[:, :, 0].copy() if c == 1 else a)
    got = aocr.data.preprocess_batch(imgs, out_w).cpu().numpy()
    assert got.shape == (len(imgs), 1, 32, out_w)
    worst = 0.0
    for i, a in enumerate(imgs):
        ref = D.scale_bilinear(D.rgb2y255(a), out_w)
        worst = max(worst, float(np.abs(got[i, 0] - ref).max()))
        np.testing.assert_array_equal(got[i, 0], ref, err_msg=f"image {i} shape {a.shape} -> 32x{out_w}")
    print(f"[parity] preprocess 32x{out_w}: {len(imgs)} images bit-identical to the restatement (max-abs {worst:.1e})")


def test_datagen_end_to_end(cuda, tmp_path):
    import aocr
    import data_oracle as D
    rng = np.random.default_rng(7)
    lines, imgs = [], {}
    for i in range(23):
        h, w = int(rng.integers(16, 64)), int(rng.integers(16, 400))
        a = rng.integers(0, 256, (h, w, 3) if i % 2 else (h, w), dtype=np.uint8)
        name = f"im{i}.npy"
        np.save(tmp_path / name, a); imgs[name] = a
        lines.append([name, "".join(rng.choice(list("abc012xyz"), int(rng.integers(1, 9))))])
    lines.insert(5, ["broken.npy", "zz"])                          # unreadable file: skipped by both
    (tmp_path / "list.txt").write_text("".join(f"{p} {l}\n" for p, l in lines))
    for force in (100, None):
        g = aocr.DataGen(str(tmp_path), "list.txt", 4.0, force_width=force)
        o = D.DataGen(lines, lambda p: imgs.get(p), 4.0, force_width=force)
        nb = 0
        while True:
            b, r = g.nextBatch(4), o.next_batch(4)
            assert (b is None) == (r is None)
            if b is None:
                break
            nb += 1
            np.testing.assert_array_equal(b[0].cpu().numpy(), r[0])
            np.testing.assert_array_equal(b[1], r[1]); np.testing.assert_array_equal(b[2], r[2])
            assert b[3] == r[3] and b[4] == r[4]
        assert nb >= 6 and g.cursor == 0
        print(f"[parity] DataGen force_width={force}: {nb} batches identical to the restatement")


def test_datagen_feeds_model_step(cuda, tmp_path):
    """A DataGen batch drops into Model.step like a reference batch (train.lua's loop)."""
    import aocr
    rng = np.random.default_rng(3)
    with open(tmp_path / "l.txt", "w") as f:
        for i in range(4):
            np.save(tmp_path / f"a{i}.npy", rng.integers(0, 256, (32, 120), dtype=np.uint8))
            f.write(f"a{i}.npy ab{i}\n")
    g = aocr.DataGen(str(tmp_path), "l.txt", 8.0)
    batch = g.nextBatch(4)
    m = aocr.Model().create(dict(encoder_num_hidden=32, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=4,
                                 max_img_w=100, max_decoder_l=8, max_beam=1, learning_rate=0.1, seed=1))
    loss, stats = m.step(batch, forward_only=False)
    assert np.isfinite(loss) and stats[0] == batch[3]
    m.shutdown()


def test_preprocess_matches_golden_fixture(cuda):
    """aocr_preprocess_lines against the committed fixture tests/golden/data_path.npz (inputs regenerated from the counter-based
    generator; expected outputs come from the file), bit for bit."""
    import importlib.util
    import aocr
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(root, "oracle", "gen_golden.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    ref = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_path.npz"))
    shapes = [(20, 37, 3), (64, 300, 3), (48, 100, 1), (7, 500, 1), (33, 129, 3)]
    for i, (h, w, c) in enumerate(shapes):
        a = np.floor(g.O.counter_uniform(g.SEED, 2000 + i, h * w * c) * 256.0).astype(np.uint8).reshape(h, w, c)
        a = np.ascontiguousarray(a[:, :, 0]) if c == 1 else a
        for force in (100, None):
            img_w = int(ref[f"img{i}:w{force}"])
            got = aocr.data.preprocess_batch([a], img_w).cpu().numpy()[0, 0]
            np.testing.assert_array_equal(got, ref[f"img{i}:out{force}"], err_msg=f"img{i} force {force}")
    assert aocr.data.str2numlist("a0z9hello42") == ref["labels"].tolist()
    print("[parity] preprocess: 10 fixture outputs bit-identical")
