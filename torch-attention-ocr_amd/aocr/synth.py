"""Deterministic synthetic weights and batches.

Counter-based generator: value i of stream s under seed is
splitmix64(splitmix64(seed ^ s*K) + i), so Python and any other host language can
reproduce the exact tensors without shipping weight files.  Used by bench.py and the
tests; the oracle carries its own copy and a test asserts the two agree.
"""
from __future__ import annotations

import math

import numpy as np

PAD, GO, EOS = 1, 2, 3          # src/train.lua:53
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def counter_uniform(seed: int, stream: int, n: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95))], dtype=np.uint64))[0]
        r = _splitmix64((base + np.arange(n, dtype=np.uint64)) & _M64)
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def counter_normal(seed: int, stream: int, n: int) -> np.ndarray:
    u = counter_uniform(seed, stream, 2 * n)
    return np.sqrt(-2.0 * np.log(np.maximum(u[0::2], 1e-300))) * np.cos(2.0 * math.pi * u[1::2])


def synth_batch(B: int, W: int, seed: int = 1234, min_len: int = 4, max_len: int = 23, force_max: bool = True,
                vocab: int = 39, H: int = 32):
    """A batch in the layout DataGen emits (src/data/data_gen.lua:100-120):
    [images (B,1,H,W) float64 0..255, targets (B,L) int32 = GO ids.. PAD.., targets_eval = ids.. EOS PAD.., num_nonzeros]."""
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
        nnz += n + 1
    return img, targets, targets_eval, nnz
