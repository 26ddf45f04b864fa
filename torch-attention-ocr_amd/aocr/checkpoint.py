"""aocr.checkpoint -- the reference's checkpoint as data.

`model:save` (src/model/model.lua:720-725) writes, with `torch.save`,
    { {cnn_model, encoder_fw, encoder_bw, decoder, output_projector}, config, global_step, optim_state }
and `model:load` (:45-80) reads it back.  `read_reference_checkpoint` parses such a file with aocr.t7 and pulls the parameter
tensors out of the five serialized nets BY STRUCTURE, not by position:

  * cnn_model (nn.Sequential, cnn.lua:9-45): the seven *SpatialConvolution* modules in order are conv1..conv7, the three
    *BatchNormalization modules follow conv3, conv5, conv7 (weight, bias, running_mean, running_var -- or the older running_std);
  * encoder_fw / encoder_bw / decoder (nn.gModule, LSTM.lua:18-128): every layer sums two Linear modules in one nn.CAddTable,
    called as `CAddTable()({i2h, h2h})` (LSTM.lua:86-88): the first parent of that node is i2h, the second h2h; the layers are
    ordered by the position of their CAddTable in the graph's topological order (each depends on the one below).  When the run
    used -prealloc the modules also carry the names memory.lua:55-66 gave them (`decoder_L2_h2h-reuse`, ...); they are
    cross-checked.  nn.LookupTable is the embedding, the nested attention gModule (LSTM.lua:130-162) holds LinearNoBias (H,H) = W_a
    and LinearNoBias (H,2H) = W_c;
  * output_projector (nn.Sequential, output_projector.lua:3-8): its nn.Linear.

PARITY UNPINNED, and stated as such: no Torch7 runs here and the reference ships no checkpoint, so this reader has only seen files
produced by aocr.t7's own writer from object trees that restate what nn / nngraph serialize ([upstream] nngraph.Node: fields
`data` {module, mapindex[i] = parent's data, ...}, `children`; nn.gModule: `forwardnodes`) -- tests/test_t7_cpu.py.

Two ways back:
  * `write_reference_checkpoint` (round 3) emits the reference's OWN layout -- the five nets as nn / nngraph object trees built the
    way cnn.lua:9-45, LSTM.lua:18-162 and output_projector.lua:3-8 build them, {nets, config, global_step, optim_state} -- so that
    the reference's `model:load` (model.lua:45-80) has a file of the shape it expects.  UNVERIFIED AGAINST TORCH7: what is checked is
    that this package's reader (written against the fixture trees of tests/t7_fixtures.py, an independent restatement) recovers every
    tensor bit for bit; the private fields nn / nngraph / cudnn.torch add to their objects beyond those the constructors of the
    un-pinned upstream packages set are not knowable here.
  * `write_flat_checkpoint`: a `.t7` holding ONE plain table of named FloatTensors in Torch7 layouts (+ BatchNorm running statistics,
    config, global_step, optim_state) that ten lines of Lua pour into a freshly created reference model -- the
    conservative route, which depends on nothing but torch.load of a plain table.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import t7
from .t7 import LuaTable, TorchObject

CONFIG_KEYS = ("dropout", "encoder_num_hidden", "encoder_num_layers", "decoder_num_hidden", "decoder_num_layers", "target_vocab_size",
               "target_embedding_size", "max_encoder_l", "max_decoder_l", "input_feed", "batch_size", "prealloc")    # model.lua:131-142


class CheckpointError(ValueError):
    pass


def _cls(obj) -> str:
    return obj.typename if isinstance(obj, TorchObject) else ""


def _leaf_modules(mod) -> List[TorchObject]:
    """modules of a container in `modules` order, containers flattened (nn.Sequential:parameters() order [upstream])."""
    mods = mod.get("modules") if isinstance(mod, TorchObject) else None
    if isinstance(mods, dict) and _cls(mod) != "nn.gModule":
        out = []
        for m in LuaTable(mods).array_part():
            out += _leaf_modules(m)
        return out
    return [mod]


def _tensor(x, what) -> np.ndarray:
    if not isinstance(x, np.ndarray):
        raise CheckpointError(f"{what}: expected a tensor, found {type(x).__name__}")
    return np.array(x, dtype=np.float32)


def _graph_nodes(g: TorchObject) -> List[dict]:
    """`data` tables of a gModule's forward nodes in topological order."""
    fn = g.get("forwardnodes")
    if not isinstance(fn, dict):
        raise CheckpointError("nn.gModule without forwardnodes")
    out = []
    for node in LuaTable(fn).array_part():
        data = node.get("data") if isinstance(node, TorchObject) else (node.get("data") if isinstance(node, dict) else None)
        if isinstance(data, dict):
            out.append(data)
    return out


def _parents(data: dict) -> List[dict]:
    mi = data.get("mapindex")
    return LuaTable(mi).array_part() if isinstance(mi, dict) else []


_NAME = re.compile(r"^(.*)_L(\d+)_(i2h|i2h-reuse|h2h-reuse)$")


def _lstm_params(g: TorchObject, prefix: str, n_layers: int, hidden: int, out: Dict[str, np.ndarray]):
    if _cls(g) != "nn.gModule":
        raise CheckpointError(f"{prefix}: expected nn.gModule, found {_cls(g) or type(g).__name__}")
    nodes = _graph_nodes(g)
    layer = 0
    for data in nodes:
        m = data.get("module")
        if _cls(m) != "nn.CAddTable":
            continue
        par = [p.get("module") for p in _parents(data)]
        if len(par) != 2 or not all(_cls(p) == "nn.Linear" for p in par):
            continue                                             # the cell-state sums (LSTM.lua:106-109) add CMulTables
        w = [_tensor(p.get("weight"), f"{prefix} Linear.weight") for p in par]
        if w[0].shape[0] != 4 * hidden or w[1].shape != (4 * hidden, hidden):
            continue
        layer += 1
        for role, p, wt in (("i2h", par[0], w[0]), ("h2h", par[1], w[1])):
            name = p.get("name")
            if isinstance(name, str):
                mt = _NAME.match(name)
                if not mt or int(mt.group(2)) != layer or mt.group(3).split("-")[0] != role:
                    raise CheckpointError(f"{prefix}: module named '{name}' sits where layer {layer} {role} is expected")
            out[f"{prefix}.l{layer}.{role}.w"] = wt
            out[f"{prefix}.l{layer}.{role}.b"] = _tensor(p.get("bias"), f"{prefix}.l{layer}.{role}.bias")
    if layer != n_layers:
        raise CheckpointError(f"{prefix}: found {layer} LSTM layers in the graph, the config says {n_layers}")
    return nodes


def read_reference_checkpoint(path_or_bytes) -> dict:
    """-> {"params": {name: float32 array in Torch7 layout}, "bn_state": {cnn.bn{3,5,7}.{rm,rv}}, "config", "global_step",
    "optim_state"} with the parameter names of aocr.Model.set_parameters."""
    ck = t7.loads(path_or_bytes) if isinstance(path_or_bytes, (bytes, bytearray)) else t7.load(path_or_bytes)
    return parse_reference_checkpoint(ck)


def parse_reference_checkpoint(ck) -> dict:
    if not isinstance(ck, dict) or not isinstance(ck.get(1), dict) or not isinstance(ck.get(2), dict):
        raise CheckpointError("not a reference checkpoint: expected {nets, config, global_step, optim_state} (model.lua:724)")
    nets = LuaTable(ck[1]).array_part()
    if len(nets) != 5:
        raise CheckpointError(f"expected 5 nets (model.lua:724), found {len(nets)}")
    config = {k: ck[2].get(k) for k in CONFIG_KEYS if k in ck[2]}
    He, Le, Ld = int(config["encoder_num_hidden"]), int(config["encoder_num_layers"]), int(config["decoder_num_layers"])
    Hd = 2 * He
    P: Dict[str, np.ndarray] = {}
    S: Dict[str, np.ndarray] = {}
    # ---- CNN
    conv_i, last_conv = 0, 0
    for m in _leaf_modules(nets[0]):
        c = _cls(m)
        if c.endswith("SpatialConvolution") or c.endswith("SpatialConvolutionMM"):
            conv_i += 1; last_conv = conv_i
            w = _tensor(m.get("weight"), f"conv{conv_i}.weight")
            if w.ndim == 2:                                      # nn.SpatialConvolutionMM keeps [Cout][Cin*kH*kW]
                w = w.reshape(int(m["nOutputPlane"]), int(m["nInputPlane"]), int(m["kH"]), int(m["kW"]))
            P[f"cnn.conv{conv_i}.w"] = w; P[f"cnn.conv{conv_i}.b"] = _tensor(m.get("bias"), f"conv{conv_i}.bias")
        elif c.endswith("BatchNormalization"):
            i = last_conv
            P[f"cnn.bn{i}.w"] = _tensor(m.get("weight"), f"bn{i}.weight"); P[f"cnn.bn{i}.b"] = _tensor(m.get("bias"), f"bn{i}.bias")
            S[f"cnn.bn{i}.rm"] = _tensor(m.get("running_mean"), f"bn{i}.running_mean")
            if isinstance(m.get("running_var"), np.ndarray):
                S[f"cnn.bn{i}.rv"] = _tensor(m.get("running_var"), f"bn{i}.running_var")
            else:                                                # older nn: running_std = 1 / sqrt(var + eps)
                std = _tensor(m.get("running_std"), f"bn{i}.running_std").astype(np.float64)
                S[f"cnn.bn{i}.rv"] = (1.0 / (std * std) - float(m.get("eps", 1e-5))).astype(np.float32)
    if conv_i != 7 or sorted(k for k in S if k.endswith(".rm")) != ["cnn.bn3.rm", "cnn.bn5.rm", "cnn.bn7.rm"]:
        raise CheckpointError(f"cnn_model: expected 7 convolutions with BatchNorm after 3, 5, 7 (cnn.lua:9-45); found {conv_i} "
                              f"and {sorted(S)}")
    # ---- recurrent nets
    _lstm_params(nets[1], "enc_fw", Le, He, P)
    _lstm_params(nets[2], "enc_bw", Le, He, P)
    dec_nodes = _lstm_params(nets[3], "dec", Ld, Hd, P)
    for data in dec_nodes:
        m = data.get("module")
        if _cls(m) == "nn.LookupTable":
            P["dec.lookup"] = _tensor(m.get("weight"), "LookupTable.weight")
        elif _cls(m) == "nn.gModule":                            # decoder_attn, LSTM.lua:112-114,130-162
            for d2 in _graph_nodes(m):
                m2 = d2.get("module")
                if _cls(m2) in ("nn.LinearNoBias", "nn.Linear") and isinstance(m2.get("weight"), np.ndarray):
                    w = _tensor(m2.get("weight"), "attention weight")
                    if w.shape == (Hd, Hd):
                        P["dec.attn.wa"] = w
                    elif w.shape == (Hd, 2 * Hd):
                        P["dec.attn.wc"] = w
    for k in ("dec.lookup", "dec.attn.wa", "dec.attn.wc"):
        if k not in P:
            raise CheckpointError(f"decoder: {k} not found in the graph")
    # ---- projector
    lin = [m for m in _leaf_modules(nets[4]) if _cls(m) == "nn.Linear"]
    if len(lin) != 1:
        raise CheckpointError("output_projector: expected exactly one nn.Linear (output_projector.lua:5)")
    P["proj.w"] = _tensor(lin[0].get("weight"), "proj.weight"); P["proj.b"] = _tensor(lin[0].get("bias"), "proj.bias")
    if P["proj.w"].shape != (int(config["target_vocab_size"]), Hd):
        raise CheckpointError(f"proj.w is {P['proj.w'].shape}, the config says ({config['target_vocab_size']}, {Hd})")
    optim = dict(ck.get(4) or {}) if isinstance(ck.get(4), dict) else {}
    return {"params": P, "bn_state": S, "config": config, "global_step": int(ck.get(3) or 0), "optim_state": optim}


# ------------------------------------------------------------------------------------------------ the reference's own layout
def _module(cls: str, **fields) -> TorchObject:
    """an nn.Module as nn.Module.__init leaves it ([upstream] nn/Module.lua: gradInput, output, _type) + its own fields."""
    f = LuaTable(gradInput=np.zeros((0,), np.float32), output=np.zeros((0,), np.float32), _type="torch.FloatTensor")
    f.update(fields)
    return TorchObject(cls, f)


def _param_fields(weight, bias=None) -> dict:
    w = np.ascontiguousarray(np.asarray(weight, np.float32))
    f = dict(weight=w, gradWeight=np.zeros_like(w))
    if bias is not None:
        b = np.ascontiguousarray(np.asarray(bias, np.float32))
        f.update(bias=b, gradBias=np.zeros_like(b))
    return f


class _Graph:
    """nngraph as data: node(module, *parents) -> nngraph.Node ([upstream] graph.Node + nngraph: `data` = {module, mapindex}, where
    mapindex[i] is the i-th parent's data table and mapindex[that table] = i; `children`; a gModule lists its nodes in forward
    (topological) order in `forwardnodes`).  Nodes are created in the order the reference's Lua creates them, which is topological."""

    def __init__(self):
        self.nodes = []

    def node(self, module, *parents):
        data = LuaTable(module=module)
        mi = LuaTable()
        for i, par in enumerate(parents, 1):
            pd = par.fields["data"]
            mi[i] = pd; mi[pd] = i
        data["mapindex"] = mi
        n = TorchObject("nngraph.Node", LuaTable(data=data, children=LuaTable(), id=len(self.nodes) + 1, visited=False))
        for par in parents:
            ch = par.fields["children"]; ch[len(ch) + 1] = n
        self.nodes.append(n)
        return n

    def gmodule(self, n_inputs: int, name: Optional[str] = None) -> TorchObject:
        f = dict(forwardnodes=LuaTable((i, n) for i, n in enumerate(self.nodes, 1)), verbose=False, nInputs=n_inputs)
        if name:
            f["name"] = name
        return _module("nn.gModule", **f)


def _lstm_net(P, prefix: str, n_layers: int, hidden: int, dropout: float, attention=False, input_feed=False, lookup=False, mem_name=None):
    """createLSTM, LSTM.lua:18-122 (node creation order = the source's)."""
    g = _Graph()
    ident = lambda: g.node(_module("nn.Identity"))
    inputs = [ident()]                                            # x, :30
    offset = 0
    if attention:
        inputs.append(ident()); offset += 1                       # context, :34
        if input_feed:
            inputs.append(ident()); offset += 1                   # prev attention output, :38
    for _ in range(n_layers):
        inputs += [ident(), ident()]                              # prev_c[L], prev_h[L], :42-45
    outputs = []
    for L in range(1, n_layers + 1):
        prev_c, prev_h = inputs[L * 2 - 1 + offset], inputs[L * 2 + offset]       # :50-51
        if L == 1:
            x = inputs[0]
            if lookup:                                            # :55-56
                x = g.node(_module("nn.LookupTable", **_param_fields(P[f"{prefix}.lookup"]), paddingValue=0), x)
            if input_feed:                                        # :59-64
                x = g.node(_module("nn.JoinTable", dimension=2), x, inputs[offset])
        else:                                                     # :66-69
            x = g.node(_module("nn.Dropout", p=float(dropout), train=True, v2=True, inplace=False), outputs[(L - 1) * 2 - 1])

        def linear(role):
            f = _param_fields(P[f"{prefix}.l{L}.{role}.w"], P[f"{prefix}.l{L}.{role}.b"])
            if mem_name:                                          # memory.lua:55-66 under -prealloc
                f["name"] = f"{mem_name}_L{L}_" + ("i2h-reuse" if role == "i2h" else "h2h-reuse")
            return _module("nn.Linear", **f)
        i2h = g.node(linear("i2h"), x)                            # :79-80
        h2h = g.node(linear("h2h"), prev_h)                       # :84-85
        sums = g.node(_module("nn.CAddTable"), i2h, h2h)          # :86-88
        resh = g.node(_module("nn.Reshape", size=np.array([4, hidden], np.int64), nelement=4 * hidden, batchMode=True), sums)   # :89
        split = g.node(_module("nn.SplitTable", dimension=2), resh)
        n = [g.node(_module("nn.SelectTable", index=i), split) for i in range(1, 5)]
        ig, fg, og = (g.node(_module("nn.Sigmoid"), n[i]) for i in range(3))      # :94-97 gate order in, forget, out
        it = g.node(_module("nn.Tanh"), n[3])
        next_c = g.node(_module("nn.CAddTable"), g.node(_module("nn.CMulTable"), fg, prev_c), g.node(_module("nn.CMulTable"), ig, it))   # :99-102
        next_h = g.node(_module("nn.CMulTable"), og, g.node(_module("nn.Tanh"), next_c))                                              # :104
        outputs += [next_c, next_h]
    if attention:                                                 # :110-118 + create_decoder_attn :124-162
        a = _Graph()
        ai = [a.node(_module("nn.Identity")), a.node(_module("nn.Identity"))]
        tt = a.node(_module("nn.LinearNoBias", **_param_fields(P[f"{prefix}.attn.wa"])), ai[0])
        mm1 = a.node(_module("nn.MM", transA=False, transB=False), ai[1], a.node(_module("nn.Replicate", nfeatures=1, dim=3, ndim=2), tt))
        sm = a.node(_module("nn.SoftMax", name="softmax_attn"), a.node(_module("nn.Sum", dimension=3), mm1))
        mm2 = a.node(_module("nn.MM", transA=False, transB=False), a.node(_module("nn.Replicate", nfeatures=1, dim=2, ndim=2), sm), ai[1])
        join = a.node(_module("nn.JoinTable", dimension=2), a.node(_module("nn.Sum", dimension=2), mm2), ai[0])
        a.node(_module("nn.Tanh"), a.node(_module("nn.LinearNoBias", **_param_fields(P[f"{prefix}.attn.wc"])), join))
        attn = g.node(a.gmodule(2, "decoder_attn"), outputs[-1], inputs[1])
        g.node(_module("nn.Dropout", p=float(dropout), train=True, v2=True, inplace=False), attn)
    return g.gmodule(len(inputs))


def build_reference_checkpoint(params: Dict[str, np.ndarray], bn_state: Dict[str, np.ndarray], config: dict, global_step: int,
                               optim_state: dict) -> LuaTable:
    """{ {cnn_model, encoder_fw, encoder_bw, decoder, output_projector}, config, global_step, optim_state } (model.lua:720-725) as a
    t7 object tree; `params` / `bn_state` in Torch7 layouts under the names of aocr.Model.get_parameters / get_bn_state."""
    P = params
    He, Le, Ld = int(config["encoder_num_hidden"]), int(config["encoder_num_layers"]), int(config["decoder_num_layers"])
    drop = float(config.get("dropout", 0.0))
    seq = [_module("nn.AddConstant", constant_scalar=-128.0, inplace=False), _module("nn.MulConstant", constant_scalar=1.0 / 128, inplace=False)]   # cnn.lua:9-10
    spec = {1: ("R", (2, 2)), 2: ("R", (2, 2)), 3: ("BR", None), 4: ("R", (1, 2)), 5: ("BR", None), 6: ("R", (1, 2)), 7: ("BR", None)}            # cnn.lua:12-42
    for i in range(1, 8):
        w = np.asarray(P[f"cnn.conv{i}.w"], np.float32)
        k, pad = w.shape[2], (1 if i < 7 else 0)
        seq.append(_module("cudnn.SpatialConvolution", **_param_fields(w, P[f"cnn.conv{i}.b"]), nOutputPlane=w.shape[0], nInputPlane=w.shape[1],
                           kH=k, kW=k, dW=1, dH=1, padW=pad, padH=pad, groups=1))
        acts, pool = spec[i]
        for c in acts:
            if c == "B":
                seq.append(_module("nn.SpatialBatchNormalization", **_param_fields(P[f"cnn.bn{i}.w"], P[f"cnn.bn{i}.b"]),
                                   running_mean=np.ascontiguousarray(np.asarray(bn_state[f"cnn.bn{i}.rm"], np.float32)),
                                   running_var=np.ascontiguousarray(np.asarray(bn_state[f"cnn.bn{i}.rv"], np.float32)),
                                   eps=1e-5, momentum=0.1, affine=True, train=True, nDim=4))
            else:
                seq.append(_module("cudnn.ReLU", inplace=True, mode="CUDNN_ACTIVATION_RELU"))
        if pool:
            seq.append(_module("cudnn.SpatialMaxPooling", kW=pool[0], kH=pool[1], dW=pool[0], dH=pool[1], padW=0, padH=0, ceil_mode=False))
    seq.append(_module("nn.View", size=t7.Storage(np.array([512, -1], np.int64)), numElements=512, numInputDims=3))      # nn.View keeps its size as a torch.LongStorage                                          # cnn.lua:44
    seq.append(_module("nn.Transpose", permutations=LuaTable({1: LuaTable({1: 2, 2: 3})})))                                                     # cnn.lua:45
    cnn = _module("nn.Sequential", modules=LuaTable((i, m) for i, m in enumerate(seq, 1)), train=True)
    pre = bool(config.get("prealloc", False))
    enc_fw = _lstm_net(P, "enc_fw", Le, He, drop, mem_name="encoder-fw" if pre else None)
    enc_bw = _lstm_net(P, "enc_bw", Le, He, drop, mem_name="encoder-bw" if pre else None)
    dec = _lstm_net(P, "dec", Ld, 2 * He, drop, attention=True, input_feed=bool(config.get("input_feed", False)), lookup=True,
                    mem_name="decoder" if pre else None)
    proj = _module("nn.Sequential", modules=LuaTable({1: _module("nn.Linear", **_param_fields(P["proj.w"], P["proj.b"])), 2: _module("nn.LogSoftMax")}),
                   train=True)                                    # output_projector.lua:3-8
    cfg = LuaTable((k, config[k]) for k in CONFIG_KEYS if k in config)
    cfg.setdefault("decoder_num_hidden", 2 * He)
    return LuaTable({1: LuaTable({1: cnn, 2: enc_fw, 3: enc_bw, 4: dec, 5: proj}), 2: cfg, 3: int(global_step), 4: LuaTable(optim_state)})


def write_reference_checkpoint(path: str, params, bn_state, config: dict, global_step: int, optim_state: dict, cuda: bool = False):
    """`model:save(path)` of the reference (model.lua:720-725) from Python.  cuda=True writes torch.CudaTensor class names, which is
    what a checkpoint saved from the reference's GPU run carries (and what a reference WITHOUT cutorch cannot read back)."""
    t7.save(path, build_reference_checkpoint(params, bn_state, config, global_step, optim_state), cuda=cuda)


def write_flat_checkpoint(path: str, params: Dict[str, np.ndarray], bn_state: Dict[str, np.ndarray], config: dict, global_step: int,
                          optim_state: dict):
    """One Lua table of named FloatTensors (Torch7 layouts) + running statistics + config/step/optimizer state."""
    tab = LuaTable()
    tab["params"] = LuaTable((k, np.ascontiguousarray(np.asarray(v, np.float32))) for k, v in params.items())
    tab["bn_state"] = LuaTable((k, np.ascontiguousarray(np.asarray(v, np.float32))) for k, v in bn_state.items())
    tab["config"] = LuaTable((k, config[k]) for k in CONFIG_KEYS if k in config)
    tab["global_step"] = int(global_step)
    tab["optim_state"] = LuaTable(optim_state)
    tab["format"] = "aocr-flat-1"
    t7.save(path, tab)


def read_flat_checkpoint(obj) -> Optional[dict]:
    if isinstance(obj, dict) and obj.get("format") == "aocr-flat-1":
        return {"params": {k: np.array(v, np.float32) for k, v in obj["params"].items()},
                "bn_state": {k: np.array(v, np.float32) for k, v in obj["bn_state"].items()},
                "config": dict(obj["config"]), "global_step": int(obj["global_step"]), "optim_state": dict(obj["optim_state"])}
    return None


def read_t7_checkpoint(path: str) -> dict:
    """either kind of `.t7`: the reference's own checkpoint or the flat table written by `write_flat_checkpoint`."""
    obj = t7.load(path)
    return read_flat_checkpoint(obj) or parse_reference_checkpoint(obj)
