"""Torch7 serialization (aocr.t7) and the checkpoint reader (aocr.checkpoint), SURVEY.md 8(f) row 3, without a GPU:
hand-assembled byte streams of the published format, round trips, and the reference's {nets, config, step, optim_state}
layout restated as an object tree (tests/t7_fixtures.py)."""
import struct

import numpy as np
import pytest


def i32(v):
    return struct.pack("<i", v)


def i64(v):
    return struct.pack("<q", v)


def f64(v):
    return struct.pack("<d", v)


def s(b):
    return i32(len(b)) + b


def test_known_byte_streams():
    from aocr import t7
    assert t7.loads(i32(0)) is None
    assert t7.loads(i32(1) + f64(2.5)) == 2.5
    assert t7.loads(i32(1) + f64(7.0)) == 7 and isinstance(t7.loads(i32(1) + f64(7.0)), int)
    assert t7.loads(i32(2) + s(b"hello")) == "hello"
    assert t7.loads(i32(5) + i32(1)) is True and t7.loads(i32(5) + i32(0)) is False
    # {10, "x", k = true}
    tab = i32(3) + i32(1) + i32(3) + i32(1) + f64(1) + i32(1) + f64(10) + i32(1) + f64(2) + i32(2) + s(b"x") + i32(2) + s(b"k") + i32(5) + i32(1)
    assert t7.loads(tab) == {1: 10, 2: "x", "k": True}
    # a 2x3 FloatTensor viewing a 10-element storage at offset 2 (1-based 3) with strides (4,1)
    storage = i32(4) + i32(2) + s(b"V 1") + s(b"torch.FloatStorage") + i64(10) + np.arange(10, dtype=np.float32).tobytes()
    tensor = i32(4) + i32(1) + s(b"V 1") + s(b"torch.FloatTensor") + i32(2) + i64(2) + i64(3) + i64(4) + i64(1) + i64(3) + storage
    a = t7.loads(tensor)
    assert a.dtype == np.float32 and a.tolist() == [[2, 3, 4], [6, 7, 8]]
    # the same tensor twice in a table: the second occurrence is only its index; a CudaTensor reads as float32; legacy class header
    two = i32(3) + i32(9) + i32(2) + i32(1) + f64(1) + tensor.replace(i32(4) + i32(1) + s(b"V 1") + s(b"torch.FloatTensor"),
                                                                      i32(4) + i32(1) + s(b"V 1") + s(b"torch.CudaTensor")) \
        + i32(1) + f64(2) + i32(4) + i32(1)
    t = t7.loads(two)
    assert t[1] is t[2] and t[1].dtype == np.float32
    legacy = i32(4) + i32(1) + s(b"nn.Identity") + i32(3) + i32(2) + i32(0)
    o = t7.loads(legacy)
    assert o.typename == "nn.Identity" and o.version == 0 and o.fields == {}
    # nn object with fields, default payload = one table
    lin = i32(4) + i32(1) + s(b"V 1") + s(b"nn.Linear") + i32(3) + i32(2) + i32(1) + i32(2) + s(b"train") + i32(5) + i32(1)
    o = t7.loads(lin)
    assert o.typename == "nn.Linear" and o["train"] is True
    with pytest.raises(t7.T7Error):
        t7.loads(i32(2) + i32(100) + b"short")
    with pytest.raises(t7.T7Error):
        t7.loads(i32(42))
    with pytest.raises(t7.T7Error):      # view larger than its storage
        t7.loads(tensor.replace(i64(10) + np.arange(10, dtype=np.float32).tobytes(), i64(5) + np.arange(5, dtype=np.float32).tobytes()))


def test_writer_emits_the_published_layout():
    from aocr import t7
    assert t7.dumps(None) == i32(0)
    assert t7.dumps(3) == i32(1) + f64(3.0)
    assert t7.dumps("ab") == i32(2) + s(b"ab")
    assert t7.dumps(True) == i32(5) + i32(1)
    assert t7.dumps([5]) == i32(3) + i32(1) + i32(1) + i32(1) + f64(1) + i32(1) + f64(5)
    a = np.array([[1, 2], [3, 4]], np.float64)
    exp = i32(4) + i32(1) + s(b"V 1") + s(b"torch.DoubleTensor") + i32(2) + i64(2) + i64(2) + i64(2) + i64(1) + i64(1) \
        + i32(4) + i32(2) + s(b"V 1") + s(b"torch.DoubleStorage") + i64(4) + a.tobytes()
    assert t7.dumps(a) == exp
    assert s(b"torch.CudaTensor") in t7.dumps(a.astype(np.float32), cuda=True)


def test_round_trip_keeps_aliases_cycles_and_types():
    from aocr import t7
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    cyc = t7.LuaTable(); cyc["self"] = cyc
    key = t7.LuaTable(tag="key")
    obj = t7.LuaTable({1: "x", 2: 3.25, 3: False, "t": a, "alias": a, "view": a[:, 1, ::2], "cyc": cyc, key: "table-keyed",
                       "long": np.arange(5), "byte": np.arange(5, dtype=np.uint8),
                       "obj": t7.TorchObject("nn.Linear", t7.LuaTable(weight=a, n=None)), "fn": t7.LuaFunction(b"\x1bLJ\x02", [1, 2])})
    o = t7.loads(t7.dumps(obj))
    assert o[1] == "x" and o[2] == 3.25 and o[3] is False
    assert np.array_equal(o["t"], a) and o["t"] is o["alias"] and o["obj"]["weight"] is o["t"]
    assert np.array_equal(o["view"], a[:, 1, ::2])
    assert o["cyc"]["self"] is o["cyc"]
    assert o["long"].dtype == np.int64 and o["byte"].dtype == np.uint8
    k = [k for k in o if isinstance(k, t7.LuaTable)]
    assert len(k) == 1 and k[0]["tag"] == "key" and o[k[0]] == "table-keyed"
    assert o["fn"].dumped == b"\x1bLJ\x02" and o["fn"].upvalues.list() == [1, 2]
    assert "n" in o["obj"].fields or True                 # nil-valued fields need not survive (Lua tables cannot hold nil)


@pytest.mark.parametrize("He,Le,Ld,feed,names,conv_mm,rstd", [(16, 1, 2, True, True, False, False), (16, 2, 3, True, False, False, False),
                                                               (8, 2, 1, False, False, True, True), (512 // 32, 1, 1, False, True, False, False)])
def test_reference_checkpoint_reader(He, Le, Ld, feed, names, conv_mm, rstd):
    from aocr import t7
    from aocr.checkpoint import read_reference_checkpoint
    from t7_fixtures import random_params, reference_checkpoint
    P, S, config = random_params(He, Le, Ld, feed, seed=He + Le + Ld)
    blob = t7.dumps(reference_checkpoint(P, S, config, global_step=4321, lr=0.025, seed=Ld, names=names, conv_mm=conv_mm, running_std=rstd),
                    cuda=True)
    ck = read_reference_checkpoint(blob)
    assert sorted(ck["params"]) == sorted(P)
    for k in P:
        assert ck["params"][k].shape == P[k].shape and np.array_equal(ck["params"][k], P[k]), k
    for k in S:
        assert np.allclose(ck["bn_state"][k], S[k], rtol=1e-5, atol=1e-6), k
    assert ck["global_step"] == 4321 and ck["optim_state"] == {"learningRate": 0.025}
    assert ck["config"]["encoder_num_hidden"] == He and ck["config"]["input_feed"] == bool(feed)


def test_same_shape_layers_are_told_apart_by_the_graph():
    """Layers >= 2 have i2h and h2h of identical shape: swap them in the fixture and the reader must follow the graph."""
    from aocr import t7
    from aocr.checkpoint import read_reference_checkpoint
    from t7_fixtures import random_params, reference_checkpoint
    P, S, config = random_params(8, 3, 3, True, seed=5)
    ck = read_reference_checkpoint(t7.dumps(reference_checkpoint(P, S, config, names=False, seed=11)))
    for L in (2, 3):
        assert not np.array_equal(P[f"enc_fw.l{L}.i2h.w"], P[f"enc_fw.l{L}.h2h.w"])
        assert np.array_equal(ck["params"][f"enc_fw.l{L}.i2h.w"], P[f"enc_fw.l{L}.i2h.w"])
        assert np.array_equal(ck["params"][f"dec.l{L}.h2h.w"], P[f"dec.l{L}.h2h.w"])


def test_reader_rejects_inconsistent_files():
    from aocr import t7
    from aocr.checkpoint import CheckpointError, read_reference_checkpoint
    from t7_fixtures import random_params, reference_checkpoint
    P, S, config = random_params(8, 2, 2, True)
    tree = reference_checkpoint(P, S, config)
    tree[2]["encoder_num_layers"] = 3                                 # the config table disagrees with the graph
    with pytest.raises(CheckpointError, match="LSTM layers"):
        read_reference_checkpoint(t7.dumps(tree))
    tree = reference_checkpoint(P, S, config, names=True)
    for n in tree[1][2]["forwardnodes"].values():                     # rename one module: the names no longer agree with the graph
        m = n["data"]["module"]
        if m.typename == "nn.Linear" and m["name"] == "encoder-fw_L1_h2h-reuse":
            m.fields["name"] = "encoder-fw_L2_h2h-reuse"
    with pytest.raises(CheckpointError, match="sits where"):
        read_reference_checkpoint(t7.dumps(tree))
    with pytest.raises(CheckpointError, match="not a reference checkpoint"):
        read_reference_checkpoint(t7.dumps({"a": 1}))


def test_flat_checkpoint_round_trip(tmp_path):
    from aocr.checkpoint import read_t7_checkpoint, write_flat_checkpoint
    from t7_fixtures import random_params
    P, S, config = random_params(8, 1, 2, True)
    path = str(tmp_path / "model.t7")
    write_flat_checkpoint(path, P, S, config, 77, {"learningRate": 0.1})
    ck = read_t7_checkpoint(path)
    assert all(np.array_equal(ck["params"][k], P[k]) for k in P) and all(np.array_equal(ck["bn_state"][k], S[k]) for k in S)
    assert ck["global_step"] == 77 and ck["optim_state"] == {"learningRate": 0.1} and ck["config"]["decoder_num_layers"] == 2


@pytest.mark.parametrize("He,Le,Ld,feed,pre", [(8, 1, 2, True, True), (16, 2, 3, False, False), (8, 3, 1, True, False)])
def test_reference_layout_writer_round_trip(tmp_path, He, Le, Ld, feed, pre):
    """aocr.checkpoint.write_reference_checkpoint: model:save's own table ({5 nets as nn / nngraph object trees, config, global_step,
    optim_state}, model.lua:720-725) from Python.  Read back by the reader (written against the independent fixture trees of
    t7_fixtures.py): every tensor bit for bit, and the structure model:load walks (model.lua:51-59: checkpoint[1] = the five nets in
    order, [2] config, [3] step, [4] optimizer state).  Unverified against Torch7 itself (no Lua here) -- said so in the module."""
    from aocr import t7
    from aocr.checkpoint import read_t7_checkpoint, write_reference_checkpoint
    from t7_fixtures import random_params
    P, S, config = random_params(He, Le, Ld, feed, seed=He + Le)
    config["prealloc"] = pre
    path = str(tmp_path / "model.t7")
    write_reference_checkpoint(path, P, S, config, 4321, {"learningRate": 0.025})
    ck = read_t7_checkpoint(path)
    assert set(ck["params"]) == set(P)
    assert all(np.array_equal(ck["params"][k], P[k]) for k in P) and all(np.array_equal(ck["bn_state"][k], S[k]) for k in S)
    assert ck["global_step"] == 4321 and ck["optim_state"] == {"learningRate": 0.025}
    assert ck["config"]["encoder_num_layers"] == Le and bool(ck["config"]["input_feed"]) == bool(feed)
    raw = t7.load(path)
    nets = raw[1]
    assert [nets[i].typename for i in range(1, 6)] == ["nn.Sequential", "nn.gModule", "nn.gModule", "nn.gModule", "nn.Sequential"]
    kinds = [m.typename for m in nets[1]["modules"].array_part()]
    assert kinds[:5] == ["nn.AddConstant", "nn.MulConstant", "cudnn.SpatialConvolution", "cudnn.ReLU", "cudnn.SpatialMaxPooling"]   # cnn.lua:9-15
    assert kinds.count("cudnn.SpatialConvolution") == 7 and kinds.count("nn.SpatialBatchNormalization") == 3 and kinds[-2:] == ["nn.View", "nn.Transpose"]
    # nn.View keeps its size as a torch.LongStorage (not a tensor): a bare storage object in the byte stream
    view = [m for m in nets[1]["modules"].array_part() if m.typename == "nn.View"][0]
    assert np.array_equal(np.asarray(view["size"]), np.array([512, -1])) and np.asarray(view["size"]).dtype == np.int64
    blob = open(path, "rb").read()
    assert b"torch.LongStorage" in blob
    pools = [m for m in nets[1]["modules"].array_part() if m.typename == "cudnn.SpatialMaxPooling"]
    assert [(p["kW"], p["kH"]) for p in pools] == [(2, 2), (2, 2), (1, 2), (1, 2)]                          # cnn.lua:15,20,29,38
    # gModule inputs: x [, context [, input feed]] + 2 per layer (LSTM.lua:30-45); the decoder ends in Dropout(attention) (:116-118)
    assert nets[2]["nInputs"] == 1 + 2 * Le and nets[4]["nInputs"] == 2 + (1 if feed else 0) + 2 * Ld
    last = nets[4]["forwardnodes"].array_part()[-1]["data"]["module"]
    assert last.typename == "nn.Dropout"
    if pre:
        names = [n["data"]["module"].get("name") for n in nets[4]["forwardnodes"].array_part() if n["data"]["module"].typename == "nn.Linear"]
        assert f"decoder_L{Ld}_h2h-reuse" in names and "decoder_L1_i2h-reuse" in names                      # memory.lua:55-66
