"""CPU oracle for the CNN -> BiLSTM -> attention-decoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package under
``torch-attention-ocr_amd/``) may import, call or link this file.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and only as the checker / the CPU timing baseline.

PARITY UNPINNED: the reference (da03/torch-Attention-OCR) is Lua/Torch7, holds
no tests, golden vectors or fixtures, and neither Lua nor Torch7 exists in the
build container or on the GPU box, so this restatement could not be checked
against outputs of the reference itself.  It restates the reference's Lua
op-by-op (citations below, relative to /root/reference) on top of stock
PyTorch CPU float64 arithmetic (the reference's CPU tensors are Double:
src/model/model.lua:55-59) and is cross-checked two independent ways:
  * the hand-rolled BPTT of src/model/model.lua:634-694 (``train_step_manual``)
    against PyTorch autograd over the same forward (``train_step_autograd``);
  * committed golden fixtures (tests/golden, made by oracle/gen_golden.py).

Facts about un-vendored Torch7 packages (nn, nngraph, cudnn.torch) that the
reference only ``require``s are marked [upstream].
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

PAD, GO, EOS = 1, 2, 3          # 1-based vocab ids, src/train.lua:53

# ----------------------------------------------------------------------------
# deterministic counter-based generator; duplicated on
# purpose in the product package (aocr/synth.py) -- tests assert equality.
# ----------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def counter_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """n doubles in [0,1): u[i] = splitmix64(splitmix64(seed, stream) + i) >> 11 * 2^-53."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))], dtype=np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64)
        r = _splitmix64((base + idx) & _M64)
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


# nn.Dropout (LSTM.lua:68-69: input of every LSTM layer above the first; :116-118: attention output), training mode only.  The
# reference draws from torch's global generator; the product and this oracle use a counter-based mask instead so that a step can be
# replayed: keep(idx) = (splitmix64(base + idx) >> 11) >= ceil(p 2^53), base = splitmix64(seed ^ stream * 0xD1342543DE82EF95),
# stream = 64 * train_step + site, idx = flat offset of the element in its [time][batch][hidden] buffer; kept values x 1/(1-p).
# Sites: decoder layer L's input -> L (2..), attention output -> 16, encoder fw / bw layer L's input -> 32 + L / 48 + L.
_DROP = None            # (p, seed, train_step) while a dropout_state context is active


class dropout_state:
    def __init__(self, p: float, seed: int, train_step: int):
        self.v = (float(p), int(seed), int(train_step))

    def __enter__(self):
        global _DROP
        self.prev = _DROP; _DROP = self.v if self.v[0] > 0.0 else None
        return self

    def __exit__(self, *a):
        global _DROP
        _DROP = self.prev


def dropout_mask(p: float, seed: int, train_step: int, site: int, offset: int, n: int) -> np.ndarray:
    """n mask values (0 or 1/(1-p), float64) for the elements offset .. offset+n-1 of a site's buffer."""
    with np.errstate(over="ignore"):
        stream = np.uint64(train_step) * np.uint64(64) + np.uint64(site)
        base = _splitmix64(np.array([np.uint64(seed) ^ (stream * np.uint64(0xD1342543DE82EF95))], dtype=np.uint64))[0]
        r = _splitmix64((base + np.uint64(offset) + np.arange(n, dtype=np.uint64)) & _M64)
    thr = np.uint64(math.ceil(p * 9007199254740992.0))
    return np.where((r >> np.uint64(11)) >= thr, 1.0 / (1.0 - p), 0.0)


def _drop(x, site: int, t: int):
    """Dropout of a (B, H) activation of time step t at a site (identity outside a dropout_state / in eval mode)."""
    if _DROP is None:
        return x
    p, seed, step = _DROP
    B, H = x.shape
    mk = dropout_mask(p, seed, step, site, t * B * H, B * H).reshape(B, H)
    return x * torch.from_numpy(mk).to(x.dtype)


def counter_normal(seed: int, stream: int, n: int) -> np.ndarray:
    u = counter_uniform(seed, stream, 2 * n)
    u1 = np.maximum(u[0::2], 1e-300)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u[1::2])


# ----------------------------------------------------------------------------
# configuration + parameter specification (Torch7 layouts)
# ----------------------------------------------------------------------------
@dataclass
class OcrConfig:
    """Hyper-parameters, src/train.lua:41-53 (reference defaults in comments)."""
    enc_hidden: int = 256          # -encoder_num_hidden (512)
    enc_layers: int = 1            # -encoder_num_layers (1)
    dec_layers: int = 2            # -decoder_num_layers (2)
    vocab: int = 39                # -target_vocab_size
    emb: int = 20                  # -target_embedding_size
    input_feed: bool = True        # -input_feed (README.md:4)
    cnn_feat: int = 512            # model.lua:84

    @property
    def dec_hidden(self) -> int:   # model.lua:88
        return 2 * self.enc_hidden


# (kind, ...) in module order of src/model/cnn.lua:9-45
CNN_LAYERS = [
    ("conv", 1, 1, 64, 3, 1), ("relu",), ("pool", 2, 2),
    ("conv", 2, 64, 128, 3, 1), ("relu",), ("pool", 2, 2),
    ("conv", 3, 128, 256, 3, 1), ("bn", 3, 256), ("relu",),
    ("conv", 4, 256, 256, 3, 1), ("relu",), ("pool", 2, 1),     # kH=2,kW=1 (cnn.lua:29)
    ("conv", 5, 256, 512, 3, 1), ("bn", 5, 512), ("relu",),
    ("conv", 6, 512, 512, 3, 1), ("relu",), ("pool", 2, 1),     # cnn.lua:38
    ("conv", 7, 512, 512, 2, 0), ("bn", 7, 512), ("relu",),
]
GROUPS = ["cnn", "enc_fw", "enc_bw", "dec", "proj"]             # model.lua:150


def param_spec(cfg: OcrConfig) -> List[Tuple[str, Tuple[int, ...], str, float]]:
    """(name, shape, init kind, init scale), in Torch7 getParameters() order
    (module order, weight then bias).  Init [upstream]: Linear / SpatialConvolution
    weight and bias U(+-1/sqrt(fan_in)); LookupTable N(0,1); BatchNorm weight
    U(0,1), bias 0."""
    spec = []
    for l in CNN_LAYERS:
        if l[0] == "conv":
            _, i, cin, cout, k, _ = l
            s = 1.0 / math.sqrt(k * k * cin)
            spec.append((f"cnn.conv{i}.w", (cout, cin, k, k), "uniform", s))
            spec.append((f"cnn.conv{i}.b", (cout,), "uniform", s))
        elif l[0] == "bn":
            _, i, c = l
            spec.append((f"cnn.bn{i}.w", (c,), "uniform01", 1.0))
            spec.append((f"cnn.bn{i}.b", (c,), "zero", 0.0))

    def lstm(prefix, in0, H, L):
        for layer in range(1, L + 1):
            insz = in0 if layer == 1 else H
            spec.append((f"{prefix}.l{layer}.i2h.w", (4 * H, insz), "uniform", 1.0 / math.sqrt(insz)))
            spec.append((f"{prefix}.l{layer}.i2h.b", (4 * H,), "uniform", 1.0 / math.sqrt(insz)))
            spec.append((f"{prefix}.l{layer}.h2h.w", (4 * H, H), "uniform", 1.0 / math.sqrt(H)))
            spec.append((f"{prefix}.l{layer}.h2h.b", (4 * H,), "uniform", 1.0 / math.sqrt(H)))

    He, Hd = cfg.enc_hidden, cfg.dec_hidden
    lstm("enc_fw", cfg.cnn_feat, He, cfg.enc_layers)            # model.lua:103
    lstm("enc_bw", cfg.cnn_feat, He, cfg.enc_layers)            # model.lua:104
    spec.append(("dec.lookup", (cfg.vocab, cfg.emb), "normal", 1.0))   # LSTM.lua:55
    lstm("dec", cfg.emb + (Hd if cfg.input_feed else 0), Hd, cfg.dec_layers)  # LSTM.lua:59-64
    spec.append(("dec.attn.wa", (Hd, Hd), "uniform", 1.0 / math.sqrt(Hd)))          # LSTM.lua:131
    spec.append(("dec.attn.wc", (Hd, 2 * Hd), "uniform", 1.0 / math.sqrt(2 * Hd)))  # LSTM.lua:155
    spec.append(("proj.w", (cfg.vocab, Hd), "uniform", 1.0 / math.sqrt(Hd)))        # output_projector.lua:5
    spec.append(("proj.b", (cfg.vocab,), "uniform", 1.0 / math.sqrt(Hd)))
    return spec


def group_of(name: str) -> int:
    return GROUPS.index(name.split(".")[0])


def init_params(cfg: OcrConfig, seed: int = 910820, dtype=torch.float64) -> Dict[str, torch.Tensor]:
    """Deterministic 'random-init' weights; stream id = position in param_spec."""
    out = {}
    for sid, (name, shape, kind, s) in enumerate(param_spec(cfg)):
        n = int(np.prod(shape))
        if kind == "uniform":
            v = (counter_uniform(seed, sid, n) * 2.0 - 1.0) * s
        elif kind == "uniform01":
            v = counter_uniform(seed, sid, n)
        elif kind == "normal":
            v = counter_normal(seed, sid, n)
        else:
            v = np.zeros(n)
        out[name] = torch.from_numpy(v.reshape(shape)).to(dtype)
    return out


def sharpen_params(P, wa: float = 80.0, proj: float = 8.0, lstm: float = 3.0, wc: float = 3.0):
    """Test regime away from random init (at U(+-1/sqrt(fan_in)) every logit is ~0.03, the attention is uniform and
    the loss is nnz * ln 39, so an absolute 1e-4 bound on logits is loose).  The SAME seeded weights with W_a (LSTM.lua:131), the
    projector (output_projector.lua:5), W_c (LSTM.lua:155) and every LSTM matrix (LSTM.lua:86-88) scaled: |logit| becomes O(1)
    (max ~4-6), the attention peaks (mean entropy 0.3-0.6 nat against ln T), gates leave their linear range.  Biases, the
    embedding and the CNN are untouched.  Not a trained model: a deterministic stand-in for one."""
    Q = dict(P)
    for k, v in P.items():
        if k == "dec.attn.wa": Q[k] = v * wa
        elif k == "dec.attn.wc": Q[k] = v * wc
        elif k == "proj.w": Q[k] = v * proj
        elif (k.startswith("enc_") or k.startswith("dec.l")) and k.endswith(".w"): Q[k] = v * lstm
    return Q


def attention_entropy(r):
    """Entropy (nat) of every attention distribution of a forward_train result: (L, B)."""
    ents = []
    for tr in r["dec_tr"]:
        a = tr[3][1][1]
        ents.append(-(a * (a + 1e-300).log()).sum(1))
    return torch.stack(ents)


def init_bn_state(dtype=torch.float64) -> Dict[str, torch.Tensor]:
    st = {}
    for l in CNN_LAYERS:
        if l[0] == "bn":
            st[f"cnn.bn{l[1]}.rm"] = torch.zeros(l[2], dtype=dtype)
            st[f"cnn.bn{l[1]}.rv"] = torch.ones(l[2], dtype=dtype)
    return st


def synth_batch(B: int, W: int, seed: int = 1234, min_len: int = 4, max_len: int = 23,
                force_max: bool = True, vocab: int = 39, H: int = 32):
    """Synthetic batch in the layout of src/data/data_gen.lua:100-120
    (SURVEY.md 8(d)): images (B,1,H,W) integer-valued 0..255; targets =
    [GO, ids.., PAD..], targets_eval = [ids.., EOS, PAD..], both (B, maxlen+1)."""
    img = np.floor(counter_uniform(seed, 1000, B * H * W) * 256.0).reshape(B, 1, H, W)
    lens = min_len + np.floor(counter_uniform(seed, 1001, B) * (max_len - min_len + 1)).astype(np.int64)
    if force_max:
        lens[0] = max_len
    Lm = int(lens.max())
    chars = 4 + np.floor(counter_uniform(seed, 1002, B * Lm) * (vocab - 3)).astype(np.int64).reshape(B, Lm)
    targets = np.full((B, Lm + 1), PAD, dtype=np.int32)
    targets_eval = np.full((B, Lm + 1), PAD, dtype=np.int32)
    nnz = 0
    for b in range(B):
        n = int(lens[b])
        targets[b, 0] = GO
        targets[b, 1:n + 1] = chars[b, :n]
        targets_eval[b, :n] = chars[b, :n]
        targets_eval[b, n] = EOS
        nnz += n + 1                     # data_gen.lua:110 (#label_list - 1)
    return img, targets, targets_eval, nnz


# ----------------------------------------------------------------------------
# optional operand rounding: the product's bf16 compute mode rounds BOTH operands of every
# contraction (conv2..conv7, every Linear, the attention context) to bf16 and accumulates in
# fp32.  `operand_rounding("bf16")` makes this restatement round the same operands (straight-
# through in the backward pass: gradients stay float64), so that ReLU masks and pool arg-max
# routes are those of the rounded forward pass and the bf16 path can be held to a tight bound.
# Default: identity (the reference's arithmetic, untouched).
# ----------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


_ROUND = [None]


class operand_rounding:
    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = _ROUND[0]; _ROUND[0] = self.mode
        return self

    def __exit__(self, *a):
        _ROUND[0] = self.prev


def _q(x):
    return _RoundBF16.apply(x) if _ROUND[0] == "bf16" else x


# ----------------------------------------------------------------------------
# forward building blocks
# ----------------------------------------------------------------------------
def cnn_forward(P, bn_state, images, training: bool, update_running: bool = True, decisions=None):
    """src/model/cnn.lua:9-45.  images (B,1,32,W) values 0..255 -> (B,T,512).
    decisions (tests only): the ReLU / max-pool DECISIONS of another forward pass imposed on this one -- {"idx<i>": LongTensor
    (B,C,Hp,Wp), the window element (scan order kh, kw) that pass selected after conv<i>, "act<i>": BoolTensor, whether its ReLU let
    the value through} for the layers it names.  With them the remaining differences between two passes are arithmetic only."""
    x = (images + (-128.0)) * (1.0 / 128)                                   # cnn.lua:9-10
    dec = decisions or {}
    last = 0
    skip_pool = False
    for l in CNN_LAYERS:
        if l[0] == "conv":
            last = l[1]
        if l[0] == "relu" and f"idx{last}" in dec:                           # ReLU + pool of this layer follow the imposed decisions (below)
            continue
        if l[0] == "pool" and f"idx{last}" in dec:
            kh, kw = l[1], l[2]
            Bq, Cq, Hq, Wq = x.shape
            win = x.reshape(Bq, Cq, Hq // kh, kh, Wq // kw, kw).permute(0, 1, 2, 4, 3, 5).reshape(Bq, Cq, Hq // kh, Wq // kw, kh * kw)
            x = torch.gather(win, 4, dec[f"idx{last}"].unsqueeze(-1)).squeeze(-1) * dec[f"act{last}"].to(x.dtype)
            continue
        if l[0] == "relu" and f"act{last}" in dec:
            x = x * dec[f"act{last}"].to(x.dtype)
            continue
        if l[0] == "conv":
            _, i, cin, cout, k, pad = l
            if i == 1:                                                   # K = 9: the product computes conv1 in fp32 in both modes
                x = F.conv2d(x, P[f"cnn.conv{i}.w"], P[f"cnn.conv{i}.b"], stride=1, padding=pad)
            else:
                x = F.conv2d(_q(x), _q(P[f"cnn.conv{i}.w"]), P[f"cnn.conv{i}.b"], stride=1, padding=pad)
        elif l[0] == "relu":
            x = F.relu(x)
        elif l[0] == "pool":
            x = F.max_pool2d(x, kernel_size=(l[1], l[2]), stride=(l[1], l[2]))   # floor mode
        elif l[0] == "bn":
            i = l[1]
            rm, rv = bn_state[f"cnn.bn{i}.rm"], bn_state[f"cnn.bn{i}.rv"]
            if training and not update_running:
                rm, rv = rm.clone(), rv.clone()
            # [upstream] nn.SpatialBatchNormalization: eps 1e-5, momentum 0.1,
            # biased var to normalise, unbiased var into running_var.
            x = F.batch_norm(x, rm, rv, P[f"cnn.bn{i}.w"], P[f"cnn.bn{i}.b"],
                             training=training, momentum=0.1, eps=1e-5)
    B, C, Hh, Ww = x.shape
    x = x.reshape(B, C, Hh * Ww).transpose(1, 2)                            # cnn.lua:44-45
    return x


@torch.no_grad()
def calibrated_bn_state(P, images, iters: int = 80):
    """BatchNorm running statistics that an evaluation-mode forward can use meaningfully (tests only): `iters` training-mode
    forwards of `images` with the momentum-0.1 update of cnn.lua:23,32,41, i.e. the running statistics converge to the batch
    statistics (0.9^80 = 2e-4 of the initial 0 / 1 left).  With the initial statistics an evaluation-mode CNN is not normalised at
    all, the LSTM gates saturate and every image decodes to the same string -- a decode test on them is blind to the image."""
    st = init_bn_state(dtype=images.dtype if images.dtype.is_floating_point else torch.float64)
    x = images.to(next(iter(P.values())).dtype)
    for _ in range(iters):
        cnn_forward(P, st, x, True, True)
    return st


def lstm_cell_fwd(x, c_prev, h_prev, Wi, bi, Wh, bh):
    """src/model/LSTM.lua:79-105; gate order [in, forget, out, g]."""
    H = c_prev.shape[1]
    z = _q(x) @ _q(Wi).t() + bi + _q(h_prev) @ _q(Wh).t() + bh
    i = torch.sigmoid(z[:, 0:H]); f = torch.sigmoid(z[:, H:2 * H])
    o = torch.sigmoid(z[:, 2 * H:3 * H]); g = torch.tanh(z[:, 3 * H:4 * H])
    c = f * c_prev + i * g
    tc = torch.tanh(c)
    h = o * tc
    return c, h, (i, f, o, g, tc)


def lstm_cell_bwd(dc_out, dh_out, cache, x, c_prev, h_prev, Wi, Wh):
    """Backward of the cell (chain rule through LSTM.lua:79-105)."""
    i, f, o, g, tc = cache
    do = dh_out * tc
    dc = dc_out + dh_out * o * (1 - tc * tc)
    di = dc * g; dg = dc * i; df = dc * c_prev; dc_prev = dc * f
    dz = torch.cat([di * i * (1 - i), df * f * (1 - f), do * o * (1 - o), dg * (1 - g * g)], dim=1)
    dx = dz @ Wi
    dh_prev = dz @ Wh
    return dx, dc_prev, dh_prev, dz


def attn_fwd(h_top, ctx, Wa, Wc):
    """src/model/LSTM.lua:124-162 (Luong 'general' attention + combine, no bias)."""
    q = _q(h_top) @ _q(Wa).t()                                    # LinearNoBias, LSTM.lua:131
    ctx = _q(ctx)
    s = torch.bmm(ctx, q.unsqueeze(2)).squeeze(2)                 # MM + Sum(3), :135-138
    a = torch.softmax(s, dim=1)                                   # :139-141
    c = torch.bmm(a.unsqueeze(1), ctx).squeeze(1)                 # :145-150
    cat = torch.cat([c, h_top], dim=1)                            # :153
    out = torch.tanh(_q(cat) @ _q(Wc).t())                        # :155
    return out, (q, a, c, cat)


def attn_bwd(dout, out, cache, h_top, ctx, Wa, Wc):
    q, a, c, cat = cache
    dpre = dout * (1 - out * out)
    dWc = dpre.t() @ cat
    dcat = dpre @ Wc
    Hd = h_top.shape[1]
    dc, dh = dcat[:, :Hd], dcat[:, Hd:].clone()
    da = torch.bmm(ctx, dc.unsqueeze(2)).squeeze(2)
    dctx = a.unsqueeze(2) * dc.unsqueeze(1)
    ds = a * (da - (a * da).sum(dim=1, keepdim=True))
    dq = torch.bmm(ds.unsqueeze(1), ctx).squeeze(1)
    dctx = dctx + ds.unsqueeze(2) * q.unsqueeze(1)
    dWa = dq.t() @ h_top
    dh = dh + dq @ Wa
    return dh, dctx, dWa, dWc


def projector_fwd(x, Wo, bo):
    """output_projector.lua:3-8: Linear + LogSoftMax.  Returns (logits, logp)."""
    logits = _q(x) @ _q(Wo).t() + bo
    return logits, torch.log_softmax(logits, dim=1)


# ----------------------------------------------------------------------------
# encoder / decoder forward (model.lua:284-316, 537-569)
# ----------------------------------------------------------------------------
def _lstm_w(P, prefix, layer):
    return (P[f"{prefix}.l{layer}.i2h.w"], P[f"{prefix}.l{layer}.i2h.b"],
            P[f"{prefix}.l{layer}.h2h.w"], P[f"{prefix}.l{layer}.h2h.b"])


def encoder_forward(P, cfg: OcrConfig, feats):
    """feats (B,T,512).  Returns context (B,T,2He) and per-direction traces."""
    B, T, _ = feats.shape
    He, Le = cfg.enc_hidden, cfg.enc_layers
    ctx_fw, ctx_bw = [None] * T, [None] * T
    traces = {}
    for prefix, order in (("enc_fw", range(T)), ("enc_bw", range(T - 1, -1, -1))):
        c = [feats.new_zeros(B, He) for _ in range(Le)]             # model.lua:293,305
        h = [feats.new_zeros(B, He) for _ in range(Le)]
        tr = {}
        for t in order:
            x = feats[:, t]
            step = []
            for L in range(1, Le + 1):
                Wi, bi, Wh, bh = _lstm_w(P, prefix, L)
                c_prev, h_prev = c[L - 1], h[L - 1]
                cn, hn, cache = lstm_cell_fwd(x, c_prev, h_prev, Wi, bi, Wh, bh)
                step.append((x, c_prev, h_prev, cache))
                c[L - 1], h[L - 1] = cn, hn
                x = _drop(hn, (32 if prefix == "enc_fw" else 48) + L + 1, t) if L < Le else hn      # LSTM.lua:68-69 (Dropout(0) = identity, S6)
            tr[t] = step
            (ctx_fw if prefix == "enc_fw" else ctx_bw)[t] = h[Le - 1]
        traces[prefix] = (tr, c[Le - 1], h[Le - 1])                  # final top-layer state
    context = torch.cat([torch.stack(ctx_fw, 1), torch.stack(ctx_bw, 1)], dim=2)   # model.lua:303,315
    return context, traces


def decoder_init_state(cfg: OcrConfig, traces, B, like, grad_through_quirk: bool = False):
    """model.lua:539-552 incl. quirk S5: with -input_feed and
    decoder_num_layers >= 2 the 'zero upper layers' loop zeroes h1(0) instead of h2(0)."""
    Hd, Ld = cfg.dec_hidden, cfg.dec_layers
    c_enc = torch.cat([traces["enc_fw"][1], traces["enc_bw"][1]], dim=1)
    h_enc = torch.cat([traces["enc_fw"][2], traces["enc_bw"][2]], dim=1)
    c = [like.new_zeros(B, Hd) for _ in range(Ld)]
    h = [like.new_zeros(B, Hd) for _ in range(Ld)]
    c[0], h[0] = c_enc, h_enc
    if cfg.input_feed and Ld >= 2:
        # value 0; optionally keep an identity gradient path (the reference's BPTT
        # still routes d h1(0) into the encoder, model.lua:667,681)
        h[0] = (h_enc - h_enc.detach()) if grad_through_quirk else like.new_zeros(B, Hd)
    return c, h


def decoder_step_fwd(P, cfg: OcrConfig, tok, ctx, feed, c, h, t: int = 0):
    """One decoder clone forward, LSTM.lua:18-122.
    tok (B,) 1-based ids.  Returns new (c, h, attn_out) and caches."""
    emb = P["dec.lookup"][tok.long() - 1]
    x = torch.cat([emb, feed], dim=1) if cfg.input_feed else emb            # LSTM.lua:59-64
    caches = []
    cn, hn = [], []
    for L in range(1, cfg.dec_layers + 1):
        Wi, bi, Wh, bh = _lstm_w(P, "dec", L)
        c2, h2, cache = lstm_cell_fwd(x, c[L - 1], h[L - 1], Wi, bi, Wh, bh)
        caches.append((x, c[L - 1], h[L - 1], cache))
        cn.append(c2); hn.append(h2)
        x = _drop(h2, L + 1, t) if L < cfg.dec_layers else h2                 # LSTM.lua:68-69
    out, acache = attn_fwd(hn[-1], ctx, P["dec.attn.wa"], P["dec.attn.wc"])
    out = _drop(out, 16, t)                                                   # LSTM.lua:116-118
    return cn, hn, out, (caches, acache)


def forward_train(P, bn_state, cfg: OcrConfig, images, targets, targets_eval, training=True,
                  grad_through_quirk=False, update_running=True, cnn_decisions=None):
    """model.lua:285-316 + 537-569 + loss of :643-647.  Returns dict of everything.  Inside a `dropout_state` context (and with
    training=True) the Dropout sites of LSTM.lua are active."""
    global _DROP
    if not training and _DROP is not None:                                   # evaluate(): nn.Dropout is the identity
        saved, _DROP = _DROP, None
        try:
            return forward_train(P, bn_state, cfg, images, targets, targets_eval, training, grad_through_quirk, update_running, cnn_decisions)
        finally:
            _DROP = saved
    B = images.shape[0]
    feats = cnn_forward(P, bn_state, images, training, update_running, cnn_decisions)
    context, traces = encoder_forward(P, cfg, feats)
    c, h = decoder_init_state(cfg, traces, B, feats, grad_through_quirk)
    feed = feats.new_zeros(B, cfg.dec_hidden)
    L = targets.shape[1]
    outs, logits_all, logp_all, dec_tr = [], [], [], []
    loss = feats.new_zeros(())
    w = torch.ones(cfg.vocab, dtype=feats.dtype); w[PAD - 1] = 0             # criterion.lua:4-5
    for t in range(L):
        c_prev, h_prev, feed_prev = c, h, feed
        c, h, out, caches = decoder_step_fwd(P, cfg, targets[:, t], context, feed, c, h, t)
        if cfg.input_feed:
            feed = out                                                        # model.lua:561-563
        logits, logp = projector_fwd(out, P["proj.w"], P["proj.b"])
        y = targets_eval[:, t].long() - 1
        nll = -(w[y] * logp[torch.arange(B), y]).sum()                        # ClassNLLCriterion, sizeAverage=false
        loss = loss + nll / B                                                 # model.lua:645
        outs.append(out); logits_all.append(logits); logp_all.append(logp)
        dec_tr.append((c_prev, h_prev, feed_prev, caches, c, h))
    return dict(feats=feats, context=context, traces=traces, outs=outs, logits=torch.stack(logits_all, 0),
                logp=torch.stack(logp_all, 0), loss=loss, dec_tr=dec_tr)


# ----------------------------------------------------------------------------
# training step: autograd version (independent check) and hand-rolled BPTT
# ----------------------------------------------------------------------------
def train_step_autograd(P, bn_state, cfg, images, targets, targets_eval, cnn_decisions=None):
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    st = {k: v.clone() for k, v in bn_state.items()}
    r = forward_train(Pg, st, cfg, images, targets, targets_eval, training=True, grad_through_quirk=True, cnn_decisions=cnn_decisions)
    r["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pg.items()}
    return r["loss"].detach(), grads, r, st


def train_step_manual(P, bn_state, cfg: OcrConfig, images, targets, targets_eval):
    """Restates model.lua:634-694: decoder BPTT t=L..1 with input-feed gradient,
    d(context) accumulation, encoder BPTT in both directions, CNN backward.
    The CNN (a plain nn.Sequential, model.lua:692) is differentiated by autograd."""
    assert _DROP is None, "the hand-rolled BPTT restates the reference at dropout = 0; use train_step_autograd inside dropout_state"
    st = {k: v.clone() for k, v in bn_state.items()}
    Pc = {k: (v.clone().requires_grad_(True) if k.startswith("cnn.") else v) for k, v in P.items()}
    with torch.enable_grad():
        feats = cnn_forward(Pc, st, images, True, True)
    fd = feats.detach()
    B, T, _ = fd.shape
    He, Hd, Le, Ld, V = cfg.enc_hidden, cfg.dec_hidden, cfg.enc_layers, cfg.dec_layers, cfg.vocab
    with torch.no_grad():
        context, traces = encoder_forward(P, cfg, fd)
        c, h = decoder_init_state(cfg, traces, B, fd)
        feed = fd.new_zeros(B, Hd)
        L = targets.shape[1]
        dec_tr, outs = [], []
        for t in range(L):
            c_prev, h_prev, feed_prev = c, h, feed
            c, h, out, caches = decoder_step_fwd(P, cfg, targets[:, t], context, feed, c, h)
            if cfg.input_feed:
                feed = out
            outs.append(out)
            dec_tr.append((c_prev, h_prev, feed_prev, caches))
        G = {k: torch.zeros_like(v) for k, v in P.items()}                   # model.lua:637-639
        dctx = torch.zeros_like(context)                                     # :640-641
        w = torch.ones(V, dtype=fd.dtype); w[PAD - 1] = 0
        loss = fd.new_zeros(())
        dc = [fd.new_zeros(B, Hd) for _ in range(Ld)]
        dh = [fd.new_zeros(B, Hd) for _ in range(Ld)]
        dfeed = fd.new_zeros(B, Hd)
        logits_all = [None] * L
        for t in range(L - 1, -1, -1):                                        # :643
            out = outs[t]
            logits, logp = projector_fwd(out, P["proj.w"], P["proj.b"])       # :644
            logits_all[t] = logits
            y = targets_eval[:, t].long() - 1
            loss = loss + (-(w[y] * logp[torch.arange(B), y]).sum()) / B      # :645
            dlogp = torch.zeros_like(logp)
            dlogp[torch.arange(B), y] = -w[y] / B                              # :646-647
            dlogits = dlogp - torch.exp(logp) * dlogp.sum(dim=1, keepdim=True)
            G["proj.w"] += dlogits.t() @ out; G["proj.b"] += dlogits.sum(0)
            dout = dfeed + dlogits @ P["proj.w"]                               # :648-649
            c_prev, h_prev, feed_prev, (caches, acache) = dec_tr[t]
            # attention block backward
            h_top = caches[-1][3][2] * caches[-1][3][4]                        # o * tanh(c) = h_top
            dh_top, dctx_t, dWa, dWc = attn_bwd(dout, out, acache, h_top, context, P["dec.attn.wa"], P["dec.attn.wc"])
            G["dec.attn.wa"] += dWa; G["dec.attn.wc"] += dWc
            dctx += dctx_t                                                     # :652-653
            dh[Ld - 1] = dh[Ld - 1] + dh_top
            dx = None
            for Lr in range(Ld, 0, -1):
                Wi, bi, Wh, bh = _lstm_w(P, "dec", Lr)
                x, cp, hp, cache = caches[Lr - 1]
                dx, dcp, dhp, dz = lstm_cell_bwd(dc[Lr - 1], dh[Lr - 1], cache, x, cp, hp, Wi, Wh)
                G[f"dec.l{Lr}.i2h.w"] += dz.t() @ x; G[f"dec.l{Lr}.i2h.b"] += dz.sum(0)
                G[f"dec.l{Lr}.h2h.w"] += dz.t() @ hp; G[f"dec.l{Lr}.h2h.b"] += dz.sum(0)
                dc[Lr - 1], dh[Lr - 1] = dcp, dhp
                if Lr > 1:
                    dh[Lr - 2] = dh[Lr - 2] + dx
            demb = dx[:, :cfg.emb]
            G["dec.lookup"].index_add_(0, targets[:, t].long() - 1, demb)
            dfeed = dx[:, cfg.emb:].clone() if cfg.input_feed else fd.new_zeros(B, Hd)   # :654-657
        # encoder BPTT, model.lua:662-690
        dfeats = torch.zeros_like(fd)
        for prefix, half, order in (("enc_fw", slice(0, He), range(T - 1, -1, -1)),
                                    ("enc_bw", slice(He, 2 * He), range(T))):
            tr = traces[prefix][0]
            dce = [fd.new_zeros(B, He) for _ in range(Le)]
            dhe = [fd.new_zeros(B, He) for _ in range(Le)]
            dce[Le - 1] = dc[0][:, half].clone()                               # :666 / :680
            dhe[Le - 1] = dh[0][:, half].clone()                               # :667 / :681 (quirk S5: passed even when h1(0) was zeroed)
            for t in order:
                dhe[Le - 1] = dhe[Le - 1] + dctx[:, t, half]                   # :670 / :684
                dx = None
                for Lr in range(Le, 0, -1):
                    Wi, bi, Wh, bh = _lstm_w(P, prefix, Lr)
                    x, cp, hp, cache = tr[t][Lr - 1]
                    dx, dcp, dhp, dz = lstm_cell_bwd(dce[Lr - 1], dhe[Lr - 1], cache, x, cp, hp, Wi, Wh)
                    G[f"{prefix}.l{Lr}.i2h.w"] += dz.t() @ x; G[f"{prefix}.l{Lr}.i2h.b"] += dz.sum(0)
                    G[f"{prefix}.l{Lr}.h2h.w"] += dz.t() @ hp; G[f"{prefix}.l{Lr}.h2h.b"] += dz.sum(0)
                    dce[Lr - 1], dhe[Lr - 1] = dcp, dhp
                    if Lr > 1:
                        dhe[Lr - 2] = dhe[Lr - 2] + dx
                dfeats[:, t] += dx                                             # :675 / :689
    feats.backward(dfeats)                                                     # :692
    for k, v in Pc.items():
        if k.startswith("cnn."):
            G[k] = v.grad if v.grad is not None else torch.zeros_like(v)
    return loss, G, dict(feats=fd, context=context, logits=torch.stack(logits_all, 0), dfeats=dfeats, dctx=dctx), st


def sgd_list(P, G, lr: float, clip: float = 5.0):
    """src/optim/optim_sgd.lua:38-95 with the unused options at 0: per group
    (cnn, enc_fw, enc_bw, dec, proj) clip ||g||_2 to 5, then w -= lr*g.
    Returns (new params, per-group (param norm, grad norm))."""
    norms = []
    newP = {}
    for gi, gname in enumerate(GROUPS):
        keys = [k for k in P if group_of(k) == gi]
        gn = math.sqrt(sum(float((G[k] ** 2).sum()) for k in keys))
        pn = math.sqrt(sum(float((P[k] ** 2).sum()) for k in keys))
        scale = (clip / gn) if gn > clip else 1.0                               # optim_sgd.lua:50-52
        for k in keys:
            newP[k] = P[k] - lr * (G[k] * scale)                                # :90
        norms.append((pn, gn))
    return newP, norms


def adadelta_list(P, G, state, rho: float = 0.9, eps: float = 1e-6, wd: float = 0.0):
    """src/optim/optim_adadelta.lua:19-62, tensor op by tensor op; `state` maps name -> {"var", "acc"} (created on first use,
    :43-47).  wd follows the intent of :37 (dfdy:add(wd, y)); the line as written indexes the gradient TABLE and would raise."""
    newP = {}
    for k in P:
        g = G[k] + wd * P[k] if wd != 0 else G[k]
        st = state.setdefault(k, {"var": torch.zeros_like(P[k]), "acc": torch.zeros_like(P[k])})
        st["var"] = st["var"] * rho + (1 - rho) * g * g                           # :48
        std = (st["var"] + eps).sqrt()                                            # :49
        delta = (st["acc"] + eps).sqrt() / std * g                                # :50
        newP[k] = P[k] - delta                                                    # :51
        st["acc"] = st["acc"] * rho + (1 - rho) * delta * delta                   # :52
    return newP


# ----------------------------------------------------------------------------
# decode (model.lua:321-536, 570-627)
# ----------------------------------------------------------------------------
def _topk_sorted(scores: torch.Tensor, k: int):
    """Descending, ties -> lowest index (our documented choice for S10)."""
    vals, idx = torch.sort(scores, dim=1, descending=True, stable=True)
    return vals[:, :k], idx[:, :k]


@torch.no_grad()
def decode_beam(P, bn_state, cfg: OcrConfig, images, targets, targets_eval, beam: int = 1, max_decoder_l: int = 50, trie=None,
                s9: str = "fixed"):
    """forward_only step: beam search (beam=1 -> greedy), back-trace, exact-match
    accuracy and the teacher-forced gold pass.  Deviations from the reference,
    both documented in DESIGN.md: S9 (t=1 parent index is computed from a 0-based
    id, i.e. always beam 1) and S10 (top-k order = sorted, ties to lowest index).
    s9="reference" replays what model.lua:402-404,516 literally does at t = 1: `raw_indices` is still the 1-based top-k index
    there, so beam_parents = floor(raw / V) + 1 is 2 (not 1) exactly when the first token is id V (= 39).  With beam > 1 that
    gathers the second of `beam` IDENTICAL replicas of the image's state (:524-533 replicate before the gather) and the parent
    recorded for t = 1 is never read by the back-trace (:573-585): no effect.  With beam = 1 the gather row is b + 1: the NEXT
    image's decoder state (an index error in Torch7 for the last row of the batch -- raised here as IndexError).  The returned
    dict carries `s9_src` = the rows gathered after the first step.
    trie: root node from oracle/dict_oracle.load_dictionary (-use_dictionary,
    model.lua:380-387,405-445,460-513) or None."""
    B = images.shape[0]
    V, Hd, Ld = cfg.vocab, cfg.dec_hidden, cfg.dec_layers
    k = min(beam, V)
    Lt = max_decoder_l
    tgt = torch.full((B, Lt), PAD, dtype=torch.int64); tgt[:, :targets.shape[1]] = torch.as_tensor(targets).long()
    tge = torch.full((B, Lt), PAD, dtype=torch.int64); tge[:, :targets_eval.shape[1]] = torch.as_tensor(targets_eval).long()
    feats = cnn_forward(P, bn_state, images, False)
    context, traces = encoder_forward(P, cfg, feats)
    c0, h0 = decoder_init_state(cfg, traces, B, feats)
    # --- beam loop, model.lua:376-536
    c = [x.clone() for x in c0]; h = [x.clone() for x in h0]
    feed = feats.new_zeros(B, Hd)
    ctx_k = context.unsqueeze(1).expand(B, k, *context.shape[1:]).reshape(B * k, *context.shape[1:])
    beam_scores = None
    tok = tgt[:, 0]
    hist_tok, hist_par = [], []
    for t in range(Lt):
        ctx_t = context if t == 0 else ctx_k
        c, h, out, _ = decoder_step_fwd(P, cfg, tok, ctx_t, feed, c, h)
        _, logp = projector_fwd(out, P["proj.w"], P["proj.b"])
        if t == 0 and trie is not None:
            from dict_oracle import select_first
            sel = [select_first(logp[b].tolist(), trie, k) for b in range(B)]          # :405-445
            cur = torch.tensor([x[0] for x in sel], dtype=torch.int64)
            beam_scores = torch.tensor([x[1] for x in sel], dtype=logp.dtype)
            nodes = [x[2] for x in sel]
            parents = torch.zeros(B, k, dtype=torch.int64)
            src = torch.arange(B).unsqueeze(1).expand(B, k).reshape(-1)
            s9_src = src.clone()
        elif t == 0:
            beam_scores, raw = _topk_sorted(logp, k)                         # :402
            cur = raw + 1
            parents = torch.zeros(B, k, dtype=torch.int64)                   # S9 fixed: parent beam 0
            src = torch.arange(B).unsqueeze(1).expand(B, k).reshape(-1)      # beam_replicate, :524-533
            if s9 == "reference" and trie is None:
                parents = cur // V                                           # :516 on the 1-based index: 1 iff the token is id V
                rep = parents + (torch.arange(B) * k).unsqueeze(1)           # row of the REPLICATED (B*k) state, :524-533
                if int(rep.max()) >= B * k:
                    raise IndexError("model.lua:524-533: index out of range (first token of the last row is id V with beam 1)")
                src = (rep // k).reshape(-1)                                 # replica r of image b is image b's state
            s9_src = src.clone()
        else:
            logp = logp.clone()
            fin = (tok == PAD) | (tok == EOS)
            logp[fin, PAD - 1] = 0.0                                         # :448-449
            total = (logp.view(B, k, V) + beam_scores.unsqueeze(2)).reshape(B, k * V)   # :450
            if trie is not None:
                from dict_oracle import select_next
                sel = [select_next(total[b].tolist(), nodes[b], k, V) for b in range(B)]   # :460-513
                raw = torch.tensor([x[1] for x in sel], dtype=torch.int64)
                beam_scores = torch.tensor([x[2] for x in sel], dtype=logp.dtype)
                nodes = [x[3] for x in sel]
            else:
                beam_scores, raw = _topk_sorted(total, k)                    # :452
            cur = raw % V + 1                                                # :456-458
            parents = raw // V                                               # :516
            src = (parents + (torch.arange(B) * k).unsqueeze(1)).reshape(-1)
        tok = cur.reshape(-1)
        hist_tok.append(cur.clone()); hist_par.append(parents.clone())
        feed = out[src] if cfg.input_feed else feats.new_zeros(B * k, Hd)
        c = [x[src] for x in c]; h = [x[src] for x in h]
    # --- back-trace, model.lua:573-585
    scores, best = beam_scores.max(dim=1)
    labels = torch.full((B, Lt), PAD, dtype=torch.int64)
    idx = best.clone()
    for t in range(Lt - 1, -1, -1):
        labels[:, t] = hist_tok[t][torch.arange(B), idx]
        idx = hist_par[t][torch.arange(B), idx]
    # exact-match accuracy, utils.lua:136-175 (compare up to first EOS)
    def cut(row):
        r = []
        for v in row.tolist():
            if v == EOS:
                break
            r.append(v)
        return r
    num_correct = sum(1 for b in range(B) if cut(labels[b]) == cut(tge[b]))
    # --- gold pass, model.lua:589-627
    c = [x.clone() for x in c0]; h = [x.clone() for x in h0]
    feed = feats.new_zeros(B, Hd)
    loss = feats.new_zeros(())
    gold = feats.new_zeros(B)
    w = torch.ones(V, dtype=feats.dtype); w[PAD - 1] = 0
    for t in range(Lt):
        c, h, out, _ = decoder_step_fwd(P, cfg, tgt[:, t], context, feed, c, h)
        _, logp = projector_fwd(out, P["proj.w"], P["proj.b"])
        y = tge[:, t] - 1
        lp = logp[torch.arange(B), y]
        loss = loss + (-(w[y] * lp).sum()) / B                               # :612
        gold = gold + torch.where(tge[:, t] != PAD, lp, torch.zeros_like(lp))  # :614-618
        if cfg.input_feed:
            feed = out
    return dict(labels=labels, scores=scores, gold_scores=gold, loss=loss * B, num_correct=num_correct,
                context=context, feats=feats, hist_tok=torch.stack(hist_tok), hist_par=torch.stack(hist_par), s9_src=s9_src)
