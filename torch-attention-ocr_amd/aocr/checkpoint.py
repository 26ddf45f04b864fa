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

`write_flat_checkpoint` is the way back: a `.t7` holding ONE plain table of named FloatTensors in Torch7 layouts (+ BatchNorm
running statistics, config, global_step, optim_state) that ten lines of Lua pour into a freshly created reference model
(INTEGRATION.md); writing the nets themselves would mean re-creating nngraph's private graph objects bit for bit, which cannot be
checked without Torch7.
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
