"""Object trees that restate what nn / nngraph serialize for the reference's five nets (cnn.lua:9-45, LSTM.lua:18-162,
output_projector.lua:3-8), built from a dict of named parameters -- test input for aocr.checkpoint.  The node order inside every
gModule is a RANDOM valid topological order, so nothing can rely on construction order."""
import random

import numpy as np

from aocr.t7 import LuaTable, TorchObject


def mod(cls, **fields):
    return TorchObject(cls, LuaTable(fields))


class G:
    """a tiny nngraph: node(module, *parents) -> nngraph.Node object."""

    def __init__(self):
        self.nodes = []

    def node(self, module, *parents):
        data = LuaTable(module=module)
        mi = LuaTable()
        for i, p in enumerate(parents, 1):
            pd = p.fields["data"]
            mi[i] = pd; mi[pd] = i                      # nngraph keeps both directions in the same table
        data["mapindex"] = mi
        n = TorchObject("nngraph.Node", LuaTable(data=data, children=LuaTable(), id=len(self.nodes) + 1, visited=False))
        n.parents = parents
        for p in parents:
            ch = p.fields["children"]; ch[len(ch) + 1] = n
        self.nodes.append(n)
        return n

    def gmodule(self, rng, name=None):
        # random topological order (Kahn with a random ready pick)
        indeg = {id(n): len(n.parents) for n in self.nodes}
        ready = [n for n in self.nodes if not n.parents]
        order = []
        while ready:
            n = ready.pop(rng.randrange(len(ready)))
            order.append(n)
            for c in LuaTable(n.fields["children"]).array_part():
                indeg[id(c)] -= 1
                if indeg[id(c)] == 0:
                    ready.append(c)
        assert len(order) == len(self.nodes)
        fields = LuaTable(forwardnodes=LuaTable((i, n) for i, n in enumerate(order, 1)), verbose=False, nInputs=1)
        if name:
            fields["name"] = name
        return TorchObject("nn.gModule", fields)


def lstm_graph(P, prefix, n_layers, hidden, rng, attention=False, input_feed=False, lookup=False, names=None):
    g = G()
    ident = lambda: g.node(mod("nn.Identity"))
    inputs = [ident()]
    offset = 0
    if attention:
        inputs.append(ident()); offset += 1
        if input_feed:
            inputs.append(ident()); offset += 1
    for _ in range(n_layers):
        inputs += [ident(), ident()]
    outputs = []
    for L in range(1, n_layers + 1):
        prev_h, prev_c = inputs[L * 2 + offset], inputs[L * 2 - 1 + offset]
        if L == 1:
            x = inputs[0]
            if lookup:
                x = g.node(mod("nn.LookupTable", weight=P[f"{prefix}.lookup"], gradWeight=np.zeros_like(P[f"{prefix}.lookup"])), x)
            if input_feed:
                x = g.node(mod("nn.JoinTable", dimension=2), x, inputs[offset])
        else:
            x = g.node(mod("nn.Dropout", p=0, train=True), outputs[(L - 1) * 2 - 1])

        def linear(role):
            f = dict(weight=P[f"{prefix}.l{L}.{role}.w"], bias=P[f"{prefix}.l{L}.{role}.b"])
            f["gradWeight"] = np.zeros_like(f["weight"]); f["gradBias"] = np.zeros_like(f["bias"])
            if names:
                f["name"] = f"{names}_L{L}_" + ("i2h-reuse" if role == "i2h" else "h2h-reuse")
                f["prealloc"] = f["name"]
            return mod("nn.Linear", **f)
        h2h = g.node(linear("h2h"), prev_h)             # created BEFORE i2h on purpose
        i2h = g.node(linear("i2h"), x)
        sums = g.node(mod("nn.CAddTable"), i2h, h2h)    # LSTM.lua:86-88: ({i2h, h2h})
        resh = g.node(mod("nn.Reshape"), sums)
        split = g.node(mod("nn.SplitTable", dimension=2), resh)
        n = [g.node(mod("nn.SelectTable", index=i), split) for i in range(1, 5)]
        ig, fg, og = (g.node(mod("nn.Sigmoid"), n[i]) for i in range(3))
        it = g.node(mod("nn.Tanh"), n[3])
        next_c = g.node(mod("nn.CAddTable"), g.node(mod("nn.CMulTable"), fg, prev_c), g.node(mod("nn.CMulTable"), ig, it))
        next_h = g.node(mod("nn.CMulTable"), og, g.node(mod("nn.Tanh"), next_c))
        outputs += [next_c, next_h]
    if attention:
        a = G()
        ai = [a.node(mod("nn.Identity")), a.node(mod("nn.Identity"))]
        tt = a.node(mod("nn.LinearNoBias", weight=P[f"{prefix}.attn.wa"]), ai[0])
        mm1 = a.node(mod("nn.MM"), ai[1], a.node(mod("nn.Replicate"), tt))
        sm = a.node(mod("nn.SoftMax", name="softmax_attn"), a.node(mod("nn.Sum"), mm1))
        mm2 = a.node(mod("nn.MM"), a.node(mod("nn.Replicate"), sm), ai[1])
        join = a.node(mod("nn.JoinTable"), a.node(mod("nn.Sum"), mm2), ai[0])
        wc = dict(weight=P[f"{prefix}.attn.wc"])
        if names:
            wc["name"] = "dec_noattn_linear"
        a.node(mod("nn.Tanh"), a.node(mod("nn.LinearNoBias", **wc), join))
        attn = g.node(a.gmodule(rng, "decoder_attn"), outputs[-1], inputs[1])
        g.node(mod("nn.Dropout", p=0), attn)
    return g.gmodule(rng)


def reference_checkpoint(P, S, config, global_step=1234, lr=0.05, seed=0, names=True, conv_mm=False, running_std=False):
    """{nets, config, global_step, optim_state} as model:save writes it (model.lua:724)."""
    rng = random.Random(seed)
    seq = []
    add = seq.append
    add(mod("nn.AddConstant", constant_scalar=-128.0)); add(mod("nn.MulConstant", constant_scalar=1.0 / 128))
    spec = {1: "RP", 2: "RP", 3: "BR", 4: "RP", 5: "BR", 6: "RP", 7: "BR"}
    for i in range(1, 8):
        w = P[f"cnn.conv{i}.w"]
        f = dict(weight=w.reshape(w.shape[0], -1) if conv_mm else w, bias=P[f"cnn.conv{i}.b"], nOutputPlane=w.shape[0], nInputPlane=w.shape[1],
                 kH=w.shape[2], kW=w.shape[3], dW=1, dH=1, padW=1, padH=1)
        add(mod("nn.SpatialConvolutionMM" if conv_mm else "cudnn.SpatialConvolution", **f))
        for c in spec[i]:
            if c == "B":
                b = dict(weight=P[f"cnn.bn{i}.w"], bias=P[f"cnn.bn{i}.b"], running_mean=S[f"cnn.bn{i}.rm"], eps=1e-5, momentum=0.1, affine=True)
                if running_std:
                    b["running_std"] = (1.0 / np.sqrt(S[f"cnn.bn{i}.rv"].astype(np.float64) + 1e-5)).astype(np.float32)
                else:
                    b["running_var"] = S[f"cnn.bn{i}.rv"]
                add(mod("nn.SpatialBatchNormalization", **b))
            elif c == "R":
                add(mod("cudnn.ReLU", inplace=True))
            else:
                add(mod("cudnn.SpatialMaxPooling", kW=2, kH=2))
    add(mod("nn.View")); add(mod("nn.Transpose"))
    cnn = mod("nn.Sequential", modules=LuaTable((i, m) for i, m in enumerate(seq, 1)), train=True)
    He, Le, Ld = config["encoder_num_hidden"], config["encoder_num_layers"], config["decoder_num_layers"]
    enc_fw = lstm_graph(P, "enc_fw", Le, He, rng, names="encoder-fw" if names else None)
    enc_bw = lstm_graph(P, "enc_bw", Le, He, rng, names="encoder-bw" if names else None)
    dec = lstm_graph(P, "dec", Ld, 2 * He, rng, attention=True, input_feed=bool(config["input_feed"]), lookup=True,
                     names="decoder" if names else None)
    proj = mod("nn.Sequential", modules=LuaTable({1: mod("nn.Linear", weight=P["proj.w"], bias=P["proj.b"]), 2: mod("nn.LogSoftMax")}))
    cfg = LuaTable(config)
    return LuaTable({1: LuaTable({1: cnn, 2: enc_fw, 3: enc_bw, 4: dec, 5: proj}), 2: cfg, 3: global_step,
                     4: LuaTable(learningRate=lr)})


def random_params(He, Le, Ld, input_feed, V=39, E=20, seed=0):
    rng = np.random.default_rng(seed)
    r = lambda *s: rng.standard_normal(s).astype(np.float32) * 0.1
    P, S = {}, {}
    chans = [(1, 64, 3), (64, 128, 3), (128, 256, 3), (256, 256, 3), (256, 512, 3), (512, 512, 3), (512, 512, 2)]
    for i, (ci, co, k) in enumerate(chans, 1):
        P[f"cnn.conv{i}.w"] = r(co, ci, k, k); P[f"cnn.conv{i}.b"] = r(co)
        if i in (3, 5, 7):
            P[f"cnn.bn{i}.w"] = np.abs(r(co)) + 0.5; P[f"cnn.bn{i}.b"] = r(co)
            S[f"cnn.bn{i}.rm"] = r(co); S[f"cnn.bn{i}.rv"] = np.abs(r(co)) + 0.7
    for pre in ("enc_fw", "enc_bw"):
        for L in range(1, Le + 1):
            P[f"{pre}.l{L}.i2h.w"] = r(4 * He, 512 if L == 1 else He); P[f"{pre}.l{L}.i2h.b"] = r(4 * He)
            P[f"{pre}.l{L}.h2h.w"] = r(4 * He, He); P[f"{pre}.l{L}.h2h.b"] = r(4 * He)
    Hd = 2 * He
    P["dec.lookup"] = r(V, E)
    for L in range(1, Ld + 1):
        P[f"dec.l{L}.i2h.w"] = r(4 * Hd, (E + (Hd if input_feed else 0)) if L == 1 else Hd); P[f"dec.l{L}.i2h.b"] = r(4 * Hd)
        P[f"dec.l{L}.h2h.w"] = r(4 * Hd, Hd); P[f"dec.l{L}.h2h.b"] = r(4 * Hd)
    P["dec.attn.wa"] = r(Hd, Hd); P["dec.attn.wc"] = r(Hd, 2 * Hd)
    P["proj.w"] = r(V, Hd); P["proj.b"] = r(V)
    config = dict(dropout=0.0, encoder_num_hidden=He, encoder_num_layers=Le, decoder_num_hidden=Hd, decoder_num_layers=Ld,
                  target_vocab_size=V, target_embedding_size=E, max_encoder_l=80, max_decoder_l=50, input_feed=bool(input_feed),
                  batch_size=8, prealloc=True)
    return P, S, config
