"""oracle/data_oracle.py -- CPU restatement of the reference's data path, numpy only.

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's cpu-baseline leg, never by the product (aocr/data.py calls the
HIP library).  PARITY UNPINNED: the reference (`src/data/data_gen.lua`, `src/utils/utils.lua`) is Lua/Torch7 and cannot run
here; its image arithmetic lives in the un-vendored, un-pinned `image` rock (torch/image, C file generic/image.c:
`image_(Main_rgb2y)`, `image_(Main_scaleBilinear)` -> `image_(Main_scaleLinear_rowcol)`), restated below from its published
algorithm.  No golden vectors exist for this path in the reference.

What is restated, with the reference line each piece follows:
  * str2numlist            utils.lua:104-118   label string -> [GO=2, ids..., EOS=3]; '0'..'9' -> 4..13, 'a'..'z' -> 14..39
                                                (any byte <= 96 goes through the digit formula, exactly as the reference does)
  * rgb2y                  data_gen.lua:70     255 * (0.299 R + 0.587 G + 0.114 B), R,G,B in [0,1]   [torch/image rgb2y]
  * target width           data_gen.lua:72-78  aspect = clamp(W/H, 0.5, max_aspect_ratio); imgW = ceil(aspect * 32); the reference
                                                then overrides imgW = 100 (line 78) -- `force_width=100` reproduces it, None removes it
  * scale_bilinear         data_gen.lua:79     image.scale(img, imgW, 32): rows (width) first, then columns (height); enlarging
                                                interpolates with scale (src-1)/(dst-1), shrinking averages the covered source span
  * DataGen.nextBatch      data_gen.lua:60-154 width buckets, a batch is emitted when a bucket reaches batch_size, leftovers are
                                                flushed bucket by bucket when the list is exhausted, then the cursor rewinds;
                                                targets = ids[:-1], targets_eval = ids[1:], padded with 1 to the longest in the batch
"""
import math

import numpy as np

IMG_H = 32
MIN_ASPECT = 0.5


def str2numlist(label):
    out = [2]
    for ch in label.encode("latin-1"):
        out.append(ch - 97 + 13 + 1 if ch > 96 else ch - 48 + 3 + 1)
    out.append(3)
    return out


def rgb2y255(img_u8):
    """img_u8: (H,W,3) or (H,W) uint8 -> (H,W) float32 in 0..255 (255 * rgb2y of the [0,1] image)."""
    a = np.asarray(img_u8)
    if a.ndim == 2:
        return a.astype(np.float32)
    r, g, b = (a[..., i].astype(np.float32) * np.float32(1.0 / 255.0) for i in range(3))
    y = np.float32(0.299) * r + np.float32(0.587) * g + np.float32(0.114) * b
    return (np.float32(255.0) * y).astype(np.float32)


def target_width(h, w, max_aspect_ratio, force_width=100):
    aspect = min(w / h, max_aspect_ratio)
    aspect = max(aspect, MIN_ASPECT)
    img_w = int(math.ceil(aspect * IMG_H))
    return force_width if force_width is not None else img_w


def _scale_line(src, dst_len):
    """One row/column of torch/image's scaleLinear_rowcol, float32 arithmetic."""
    src = np.asarray(src, np.float32)
    n = src.shape[0]
    f32 = np.float32
    if dst_len == n:
        return src.copy()
    dst = np.empty(dst_len, np.float32)
    if dst_len > n:
        if n == 1:
            dst[:] = src[0]
            return dst
        scale = f32(n - 1) / f32(dst_len - 1)
        for di in range(dst_len - 1):
            sf = f32(di) * scale
            si = int(sf)
            fr = f32(sf - f32(si))
            dst[di] = (f32(1) - fr) * src[si] + fr * src[si + 1]
        dst[dst_len - 1] = src[n - 1]
        return dst
    scale = f32(n) / f32(dst_len)
    si0_i, si0_f = 0, f32(0)
    for di in range(dst_len):
        s1 = f32(di + 1) * scale
        si1_i = int(s1)
        si1_f = f32(s1 - f32(si1_i))
        acc = (f32(1) - si0_f) * src[si0_i]
        cnt = f32(1) - si0_f
        for si in range(si0_i + 1, si1_i):
            acc = f32(acc + src[si]); cnt = f32(cnt + f32(1))
        if si1_i < n:
            acc = f32(acc + si1_f * src[si1_i]); cnt = f32(cnt + si1_f)
        dst[di] = acc / cnt
        si0_i, si0_f = si1_i, si1_f
    return dst


def scale_bilinear(img, dst_w, dst_h=IMG_H):
    """(H,W) float32 -> (dst_h,dst_w): every row to dst_w first, then every column to dst_h."""
    img = np.asarray(img, np.float32)
    tmp = np.stack([_scale_line(img[j], dst_w) for j in range(img.shape[0])], 0)
    return np.stack([_scale_line(tmp[:, i], dst_h) for i in range(dst_w)], 1)


def preprocess(img_u8, max_aspect_ratio, force_width=100):
    g = rgb2y255(img_u8)
    return scale_bilinear(g, target_width(g.shape[0], g.shape[1], max_aspect_ratio, force_width))


class DataGen:
    """Restatement of data_gen.lua's DataGen over in-memory (image, label) pairs; `loader(path)` returns an image array or
    None (a failed image.load is skipped, data_gen.lua:66-67,84-86)."""

    def __init__(self, lines, loader, max_aspect_ratio, force_width=100):
        self.lines = [list(x) for x in lines]          # [path, label]
        self.loader = loader
        self.max_aspect_ratio = max_aspect_ratio
        self.force_width = force_width
        self.cursor = 0
        self.buffer = {}                                # insertion-ordered like the Lua table walk of `next(self.buffer)`
        self.cache = {}

    def size(self):
        return len(self.lines)

    def shuffle(self, rng):
        """utils.lua:13-20 (Fisher-Yates from the top); rng(n) -> integer in 1..n replaces Lua's math.random."""
        counter = len(self.lines)
        while counter > 1:
            index = rng(counter)
            self.lines[index - 1], self.lines[counter - 1] = self.lines[counter - 1], self.lines[index - 1]
            counter -= 1

    def _emit(self, img_w):
        items = self.buffer.pop(img_w)
        images = np.stack([it[0] for it in items], 0)[:, None, :, :].astype(np.float32)
        max_len = max(len(it[1]) for it in items)
        targets = np.ones((len(items), max_len - 1), np.int32)
        targets_eval = np.ones((len(items), max_len - 1), np.int32)
        nnz = 0
        for i, it in enumerate(items):
            ids = it[1]
            nnz += len(ids) - 1
            targets[i, :len(ids) - 1] = ids[:-1]
            targets_eval[i, :len(ids) - 1] = ids[1:]
        return [images, targets, targets_eval, nnz, [it[2] for it in items]]

    def next_batch(self, batch_size):
        while self.cursor < len(self.lines):
            path, label = self.lines[self.cursor][0], self.lines[self.cursor][1]
            if self.cursor not in self.cache:
                img = self.loader(path)
                if img is not None:
                    self.cache[self.cursor] = (preprocess(img, self.max_aspect_ratio, self.force_width), str2numlist(label))
            entry = self.cache.get(self.cursor)
            self.cursor += 1
            if entry is None:
                continue
            img_w = entry[0].shape[1]
            self.buffer.setdefault(img_w, []).append((entry[0], entry[1], path))
            if len(self.buffer[img_w]) == batch_size:
                return self._emit(img_w)
        if not self.buffer:
            self.cursor = 0
            return None
        return self._emit(next(iter(self.buffer)))
