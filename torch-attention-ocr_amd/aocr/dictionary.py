"""aocr.dictionary -- the decoding dictionary (`-use_dictionary`) and the word scoring of the reference, host side.

Mirrors `loadDictionary` (src/utils/utils.lua:177-218), `string.levenshtein` (:55-94) and `evalWordErrRate` (:136-175).  The
reference keeps the trie as nested `tds.Hash` tables and walks them in Lua for every image, beam and candidate of every decode
step (src/model/model.lua:405-445, 460-513); here the trie is flattened once into three device-resident arrays and the
admissibility test runs inside the selection kernel (`project_select_kernel` / `beam_select_kernel`, csrc/ops_misc.hip):

    child_mask[n]  uint64   bit v-1 set <=> node n has a child for vocab id v (1-based, v <= 64)
    child_base[n]  int32    index of node n's first child in `child`
    child[...]     int32    child node ids, ascending v per node

Node 0 is the start symbol's node (`trie[2]`).  The edit distance runs on device too (`aocr_edit_distance`).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import numpy as np
import torch

from ._lib import TrieDesc, check, lib, ptr

PAD, GO, EOS = 1, 2, 3
MAX_VOCAB = 64


def char_id(ch: int) -> int:
    """utils.lua:202-207: bytes above 96 go through the letter formula, everything else through the digit formula."""
    return ch - 97 + 13 + 1 if ch > 96 else ch - 48 + 3 + 1


class Trie:
    """Flat dictionary trie.  `mask`, `base`, `child` are numpy arrays; `.to(device)` uploads them once."""

    def __init__(self, mask: np.ndarray, base: np.ndarray, child: np.ndarray, n_words: int = 0):
        self.mask = np.ascontiguousarray(mask, dtype=np.uint64)
        self.base = np.ascontiguousarray(base, dtype=np.int32)
        self.child = np.ascontiguousarray(child, dtype=np.int32)
        self.n_words = n_words
        self._dev = None

    @property
    def n_nodes(self) -> int:
        return int(self.mask.shape[0])

    @property
    def n_edges(self) -> int:
        return int(self.child.shape[0])

    def next(self, node: int, v: int) -> Optional[int]:
        """child of `node` for vocab id v, or None (the `trie_locations[b][vocab_id] ~= nil` test, model.lua:413)."""
        if not 1 <= v <= MAX_VOCAB:
            return None
        m = int(self.mask[node])
        if not (m >> (v - 1)) & 1:
            return None
        return int(self.child[int(self.base[node]) + bin(m & ((1 << (v - 1)) - 1)).count("1")])

    def walk(self, ids: Iterable[int], node: int = 0) -> Optional[int]:
        for v in ids:
            node = self.next(node, int(v))
            if node is None:
                return None
        return node

    def contains(self, word: str) -> bool:
        """True when `word` followed by EOS is a path from the root."""
        return self.walk([char_id(c) for c in word.encode("latin-1")] + [EOS]) is not None

    def to(self, device) -> "Trie":
        device = torch.device(device)
        mask = torch.from_numpy(self.mask.view(np.int64)).to(device)
        base = torch.from_numpy(self.base).to(device)
        child = torch.from_numpy(self.child if self.n_edges else np.zeros(1, np.int32)).to(device)
        self._dev = (mask, base, child)
        return self

    def desc(self) -> TrieDesc:
        """`aocr_trie` over the uploaded arrays (keep this Trie alive while the descriptor is in use)."""
        if self._dev is None:
            raise RuntimeError("Trie.to(device) has not been called")
        mask, base, child = self._dev
        return TrieDesc(ptr(mask), ptr(base), ptr(child), self.n_nodes, self.n_edges)


def build_trie(words: Iterable[str], allow_digit_prefix: bool = False) -> Trie:
    """loadDictionary's insertion loop (utils.lua:186-216) into node-indexed child tables, then flattened.

    -allow_digit_prefix: the root's EOS child and its ten digit children are the root itself (:192-198; the reference re-assigns
    them before every word, so a word's leading digits never leave the root)."""
    nodes: List[dict] = [{}]
    if allow_digit_prefix:
        nodes[0][EOS] = 0
        for v in range(4, 14):
            nodes[0][v] = 0
    n_words = 0
    for line in words:
        n_words += 1
        node = 0
        for ch in line.strip().encode("latin-1"):
            v = char_id(ch)
            nxt = nodes[node].get(v)
            if nxt is None:
                nxt = len(nodes); nodes.append({}); nodes[node][v] = nxt
            node = nxt
        if EOS not in nodes[node]:
            nodes[node][EOS] = len(nodes); nodes.append({})
    n = len(nodes)
    mask = np.zeros(n, np.uint64); base = np.zeros(n, np.int32)
    child: List[int] = []
    for i, tab in enumerate(nodes):
        base[i] = len(child)
        m = 0
        for v in sorted(tab):
            if 1 <= v <= MAX_VOCAB:                 # ids outside the vocabulary can never be proposed by the decoder
                m |= 1 << (v - 1); child.append(tab[v])
        mask[i] = m
    return Trie(mask, base, np.asarray(child, np.int32), n_words)


def load_dictionary(dictionary_path: str, allow_digit_prefix: bool = False, device=None) -> Trie:
    """loadDictionary(dictionary_path, allow_digit_prefix), utils.lua:177-218: one word per line."""
    try:
        f = open(dictionary_path, "r", encoding="latin-1")
    except OSError as e:
        raise FileNotFoundError(f"Error: Data file {dictionary_path} not found") from e       # utils.lua:179-182
    with f:
        trie = build_trie(f, allow_digit_prefix)
    return trie.to(device) if device is not None else trie


def levenshtein(a, b) -> int:
    """string.levenshtein (utils.lua:55-94) on two strings or id sequences; host version for single pairs."""
    a, b = list(a), list(b)
    if not a:
        return len(b)
    if not b:
        return len(a)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def edit_distance_device(labels: torch.Tensor, targets: torch.Tensor, stream=None):
    """Per-row Levenshtein distance of two (B,L) int32 device tensors cut at their first EOS, and the cut target lengths
    (`aocr_edit_distance`).  Enqueues only."""
    assert labels.shape == targets.shape and labels.dtype == torch.int32 and targets.dtype == torch.int32
    labels, targets = labels.contiguous(), targets.contiguous()
    B, L = labels.shape
    dist = torch.empty(B, dtype=torch.int32, device=labels.device)
    tlen = torch.empty(B, dtype=torch.int32, device=labels.device)
    s = stream if stream is not None else torch.cuda.current_stream(labels.device).cuda_stream
    s = s if isinstance(s, C.c_void_p) else C.c_void_p(s)
    check(lib.aocr_edit_distance(s, ptr(labels), ptr(targets), B, L, ptr(dist), ptr(tlen)), "aocr_edit_distance")
    return dist, tlen
