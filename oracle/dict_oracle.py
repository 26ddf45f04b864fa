"""oracle/dict_oracle.py -- CPU restatement of the reference's dictionary-constrained decoding and word scoring
(SURVEY.md 8(f) row 2), plain Python.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by oracle_torch.decode_beam when a trie is passed), never by the product --
the product walks a flat device-resident trie inside its selection kernels (csrc/ops_misc.hip).  PARITY UNPINNED: the
reference is Lua/Torch7, cannot run here and holds no vectors for this path; this file restates its table walking line by line.

What is restated, with the reference lines:
  * load_dictionary       utils.lua:177-218   nested tables keyed by vocab id under the start symbol's node trie[2]; every word ends
                                               in a child 3 (EOS); -allow_digit_prefix makes root[3] and root[4..13] the root itself
                                               (re-assigned before every word, so a leading digit of a word never leaves the root)
  * select_first          model.lua:405-445   step 1: walk the classes by descending log-probability, keep those the root continues
  * select_next           model.lua:460-513   later steps: walk beam*V candidates by descending total score; PAD (id 1) is always
                                               admissible (:469) and keeps the beam's node (:502-503)
  * levenshtein           utils.lua:55-94
  * eval_word_err_rate    utils.lua:136-175   (+ the edit-distance accuracy of the commented line :172 / README.md:11)

Deviations, all on paths the reference cannot execute: when fewer than beam_size candidates are admissible the reference fills
the remaining beams with the best admissible one at step 1 (:419-433); its copy of that rule for later steps (:477-497) reads an
undefined variable and calls :floor() on a number -- and is unreachable anyway, because the beam_size PAD
candidates are always admissible.  Ties in the descending sort go to the lowest index (torch.sort's order among ties is unspecified).
"""
PAD, GO, EOS = 1, 2, 3


def char_id(ch: int) -> int:
    """utils.lua:202-207 (same rule as str2numlist, utils.lua:108-112)."""
    return ch - 97 + 13 + 1 if ch > 96 else ch - 48 + 3 + 1


def load_dictionary(words, allow_digit_prefix=False):
    """words: iterable of lines (str).  Returns the root node trie[2] as nested dicts {vocab_id: node}."""
    root = {}
    for line in words:
        s = line.strip()                                           # trim(), utils.lua:189
        node = root
        if allow_digit_prefix:
            node[EOS] = root                                       # :193 "allow output nothing"
            for l in range(48, 58):
                node[l - 48 + 3 + 1] = root                        # :194-197
        for ch in s.encode("latin-1"):
            v = char_id(ch)
            if v not in node:
                node[v] = {}
            node = node[v]
        if EOS not in node:
            node[EOS] = {}
    return root


def select_first(logp_row, root, beam_size):
    """logp_row: V log-probabilities of one image at step 1.  Returns (tokens, scores, nodes) of the beam_size beams."""
    V = len(logp_row)
    order = sorted(range(V), key=lambda i: (-logp_row[i], i))
    toks, scores = [], []
    for i in order:
        if len(toks) == beam_size:
            break
        if (i + 1) in root:
            toks.append(i + 1); scores.append(logp_row[i])
    if len(toks) < beam_size:                                      # :419-433
        first = next((i for i in order if (i + 1) in root), None)
        if first is None:
            raise ValueError("the dictionary admits no first token")
        while len(toks) < beam_size:
            toks.append(first + 1); scores.append(logp_row[first])
    return toks, scores, [root[v] for v in toks]


def select_next(total_row, nodes, beam_size, V):
    """total_row: beam_size*V total scores of one image (PAD already zero-cost on finished beams, beam scores added);
    nodes: the trie node of each input beam.  Returns (tokens, raw 0-based candidate indices, scores, new nodes)."""
    order = sorted(range(len(total_row)), key=lambda i: (-total_row[i], i))
    toks, raws, scores = [], [], []
    for c in order:
        if len(toks) == beam_size:
            break
        v, beam = c % V + 1, c // V
        if v == PAD or v in nodes[beam]:                           # :469
            toks.append(v); raws.append(c); scores.append(total_row[c])
    assert len(toks) == beam_size                                  # the PAD candidate of every beam is admissible
    new_nodes = [nodes[c // V] if v == PAD else nodes[c // V][v] for v, c in zip(toks, raws)]   # :499-507
    return toks, raws, scores, new_nodes


def levenshtein(a, b):
    """string.levenshtein, utils.lua:55-94, on any two sequences."""
    a, b = list(a), list(b)
    if len(a) == 0:
        return len(b)
    if len(b) == 0:
        return len(a)
    if a == b:
        return 0
    m = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
    for i in range(len(a) + 1):
        m[i][0] = i
    for j in range(len(b) + 1):
        m[0][j] = j
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            cost = 0 if a[i - 1] == b[j - 1] else 1
            m[i][j] = min(m[i - 1][j] + 1, m[i][j - 1] + 1, m[i - 1][j - 1] + cost)
    return m[len(a)][len(b)]


def numlist2str(ids):
    """utils.lua:120-134."""
    return "".join(chr(v - 1 - 13 + 97) if v > 13 else chr(v - 1 - 3 + 48) for v in ids)


def _cut(row):
    out = []
    for v in row:
        if int(v) == EOS:
            break
        out.append(int(v))
    return out


def eval_word_err_rate(labels, target_labels):
    """utils.lua:136-175.  Returns (word errors, predicted strings, gold strings, per-row edit distance, per-row target length)."""
    werr, pred, gold, dist, tlen = 0.0, [], [], [], []
    for p_row, g_row in zip(labels, target_labels):
        p, g = numlist2str(_cut(p_row)), numlist2str(_cut(g_row))
        d = levenshtein(p, g)
        pred.append(p); gold.append(g); dist.append(d); tlen.append(len(g))
        if d != 0:
            werr += 1
    return werr, pred, gold, dist, tlen
