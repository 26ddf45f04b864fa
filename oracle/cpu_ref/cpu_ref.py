"""ctypes binding of oracle/cpu_ref/libaocr_cpu_ref.so (the C++/OpenMP restatement of the reference's CPU path).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__, bench.py's cpu_baseline leg).  Parameters travel as ONE flat vector in the
order of oracle_torch.param_spec (Torch7 getParameters() order, Torch7 layouts)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libaocr_cpu_ref.so")


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("img_h", "enc_hidden", "enc_layers", "dec_layers", "vocab", "emb", "input_feed")]


def build():
    subprocess.check_call(["make", "-C", HERE, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.aocr_cpu_ref_param_count.restype = C.c_int64
    return _lib


def make_cfg(enc_hidden, enc_layers, dec_layers, input_feed, img_h=32, vocab=39, emb=20):
    return Cfg(img_h, enc_hidden, enc_layers, dec_layers, vocab, emb, int(bool(input_feed)))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def flatten(named, names):
    return np.concatenate([np.asarray(named[k], dtype=np.float64).reshape(-1) for k in names])


def unflatten(flat, spec):
    out, o = {}, 0
    for name, shape in spec:
        n = int(np.prod(shape)); out[name] = flat[o:o + n].reshape(shape).copy(); o += n
    return out


def train_step(cfg, params, bn_state, images, tgt, tge, dtype=np.float64):
    """-> dict(loss (sum over the batch = loss*batch_size of model.lua:701), logits (L,B,V), grads (flat), bn_state (updated),
    context (B,T,2He), feats (B,T,512))."""
    L_ = lib(); sfx = "f64" if dtype == np.float64 else "f32"
    B, _, H, W = images.shape; L = tgt.shape[1]
    n = L_.aocr_cpu_ref_param_count(C.byref(cfg)); assert params.size == n, (params.size, n)
    p = np.ascontiguousarray(params, dtype=dtype); bn = np.ascontiguousarray(bn_state, dtype=dtype).copy()
    img = np.ascontiguousarray(images, dtype=dtype); tg = np.ascontiguousarray(tgt, dtype=np.int32); te = np.ascontiguousarray(tge, dtype=np.int32)
    Ho = H // 16 - 1; T = Ho * (W // 4 - 1); Hd = 2 * cfg.enc_hidden
    loss = np.zeros(1, dtype); logits = np.zeros((L, B, cfg.vocab), dtype); grads = np.zeros(n, dtype)
    ctx = np.zeros((B, T, Hd), dtype); feats = np.zeros((B, T, 512), dtype)
    rc = getattr(L_, "aocr_cpu_ref_train_step_" + sfx)(C.byref(cfg), _p(p), _p(bn), _p(img), _p(tg), _p(te), B, W, L, _p(loss), _p(logits), _p(grads),
                                                       _p(ctx), _p(feats))
    assert rc == 0
    return dict(loss=float(loss[0]), logits=logits, grads=grads, bn_state=bn, context=ctx, feats=feats)


def sgd(cfg, params, grads, lr, clip, dtype=np.float64):
    L_ = lib(); sfx = "f64" if dtype == np.float64 else "f32"
    p = np.ascontiguousarray(params, dtype=dtype).copy(); g = np.ascontiguousarray(grads, dtype=dtype).copy(); norms = np.zeros(10, dtype)
    ct = C.c_double if dtype == np.float64 else C.c_float
    assert getattr(L_, "aocr_cpu_ref_sgd_" + sfx)(C.byref(cfg), _p(p), _p(g), ct(lr), ct(clip), _p(norms)) == 0
    return p, norms.reshape(5, 2)


def decode(cfg, params, bn_state, images, tgt, tge, beam, max_decoder_l, dtype=np.float64):
    L_ = lib(); sfx = "f64" if dtype == np.float64 else "f32"
    B, _, H, W = images.shape; L = tgt.shape[1]
    p = np.ascontiguousarray(params, dtype=dtype); bn = np.ascontiguousarray(bn_state, dtype=dtype)
    img = np.ascontiguousarray(images, dtype=dtype); tg = np.ascontiguousarray(tgt, dtype=np.int32); te = np.ascontiguousarray(tge, dtype=np.int32)
    labels = np.zeros((B, max_decoder_l), np.int32); scores = np.zeros(B, dtype); gold = np.zeros(B, dtype); loss = np.zeros(1, dtype)
    assert getattr(L_, "aocr_cpu_ref_decode_" + sfx)(C.byref(cfg), _p(p), _p(bn), _p(img), _p(tg), _p(te), B, W, L, beam, max_decoder_l, _p(labels),
                                                     _p(scores), _p(gold), _p(loss)) == 0
    return dict(labels=labels, scores=scores, gold_scores=gold, loss=float(loss[0]))


def threads():
    return lib().aocr_cpu_ref_threads()
