"""Dictionary / word-scoring host logic without a GPU: the flat trie the product uploads against the
oracle's nested tables (utils.lua:177-218), and the Levenshtein restatement against known answers."""
import random

import numpy as np
import pytest


def _words(rng, n, alphabet="abcdefghijklmnopqrstuvwxyz0123456789", lo=1, hi=7):
    return ["".join(rng.choice(alphabet) for _ in range(rng.randint(lo, hi))) for _ in range(n)]


def _same_structure(trie, root):
    """simultaneous walk of the flat trie and the nested dicts; cycles (digit prefix) are cut by pairing node ids."""
    seen = {}
    stack = [(0, root)]
    while stack:
        n, d = stack.pop()
        if n in seen:
            assert seen[n] is d
            continue
        seen[n] = d
        kids = {v for v in range(1, 65) if trie.next(n, v) is not None}
        assert kids == {v for v in d if 1 <= v <= 64}, (n, kids, sorted(d))
        for v in kids:
            stack.append((trie.next(n, v), d[v]))
    return len(seen)


@pytest.mark.parametrize("digit_prefix", [False, True])
def test_flat_trie_matches_oracle_tables(digit_prefix):
    import dict_oracle as DO
    from aocr.dictionary import build_trie
    rng = random.Random(7)
    words = _words(rng, 300) + ["", "a", "ab", "abc", "7up", "42", "zzzzzzzz", " padded  "]
    trie = build_trie(words, digit_prefix)
    root = DO.load_dictionary(words, digit_prefix)
    reached = _same_structure(trie, root)
    assert reached == trie.n_nodes                     # every flat node is reachable: nothing was dropped or duplicated
    for w in ("a", "ab", "abc", "zzzzzzzz", "padded"):
        assert trie.contains(w)
    assert not trie.contains("abd")
    assert trie.contains("99abc") == digit_prefix      # utils.lua:192-198
    assert trie.mask.dtype == np.uint64 and trie.base.dtype == np.int32 and trie.child.dtype == np.int32
    assert int(sum(bin(int(m)).count("1") for m in trie.mask)) == trie.n_edges


def test_out_of_vocabulary_bytes_are_kept_like_the_reference():
    """Upper-case and punctuation go through the digit formula (utils.lua:205-207): 'A' (65) -> id 21, '{' (123) -> id 40."""
    import dict_oracle as DO
    from aocr.dictionary import build_trie, char_id
    assert char_id(ord("a")) == 14 and char_id(ord("z")) == 39 and char_id(ord("0")) == 4 and char_id(ord("9")) == 13
    assert char_id(ord("A")) == 21 and char_id(ord("{")) == 40 and DO.char_id(ord("A")) == 21
    trie = build_trie(["Ab", "{x"])
    assert trie.walk([21, 15, 3]) is not None and trie.walk([40, 37, 3]) is not None
    t2 = build_trie([" "])                             # trimmed to the empty word: EOS directly under the root
    assert t2.walk([3]) is not None and t2.n_nodes == 2


def test_levenshtein():
    import dict_oracle as DO
    from aocr.dictionary import levenshtein
    known = [("kitten", "sitting", 3), ("", "abc", 3), ("abc", "", 3), ("flaw", "lawn", 2), ("same", "same", 0), ("a", "b", 1),
             ("intention", "execution", 5)]
    for a, b, d in known:
        assert DO.levenshtein(a, b) == d and levenshtein(a, b) == d
    rng = random.Random(3)
    for _ in range(200):
        a, b = _words(rng, 2, "abc", 0, 9)
        assert levenshtein(a, b) == DO.levenshtein(a, b) == DO.levenshtein(b, a)


def test_oracle_selection_rules():
    """model.lua:405-445 / 460-513 on a hand-made case."""
    import dict_oracle as DO
    root = DO.load_dictionary(["ab", "b"])
    V = 39
    lp = [-10.0] * V
    lp[14 - 1], lp[15 - 1], lp[20 - 1] = -1.0, -2.0, -0.5          # a, b admissible; 'g' is the best class but not in the trie
    toks, sc, nodes = DO.select_first(lp, root, 3)
    assert toks == [14, 15, 14] and sc == [-1.0, -2.0, -1.0]       # two admissible classes, the best one fills beam 3 (:419-433)
    total = [-50.0] * (3 * V)
    total[0 * V + 15 - 1] = -1.5        # beam 0 (after 'a'): 'b' admissible
    total[1 * V + 3 - 1] = -2.5         # beam 1 (after 'b'): EOS admissible
    total[1 * V + 14 - 1] = -0.1        # beam 1: 'a' not admissible
    total[2 * V + 0] = -3.0             # PAD always admissible (:469)
    toks, raws, sc, new = DO.select_next(total, nodes, 3, V)
    assert toks == [15, 3, 1] and raws == [14, V + 2, 2 * V] and sc == [-1.5, -2.5, -3.0]
    assert new[2] is nodes[2] and new[1] == {} and 3 in new[0]


def test_eval_word_err_rate_oracle_vs_host():
    import dict_oracle as DO
    from aocr import eval_word_err_rate
    rng = np.random.default_rng(0)
    lab = rng.integers(1, 40, size=(40, 12)); tgt = lab.copy()
    tgt[::3, 4] = 3; lab[::3, 4] = 3                    # same prefix, different tails after EOS: still correct
    lab[1::3, 2] = (lab[1::3, 2] % 36) + 4               # one substitution (never EOS)
    w, pred, gold = eval_word_err_rate(lab, tgt, True)
    wo, po, go, dist, tlen = DO.eval_word_err_rate(lab, tgt)
    assert w == wo and pred == po and gold == go
    assert all((d == 0) == (p == g) for d, p, g in zip(dist, po, go))
