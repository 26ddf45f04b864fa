"""Host-side mirror of the reference's ``DataGen`` (src/data/data_gen.lua) over the HIP data-path entry point.

Same surface as the Lua class -- ``DataGen(data_base_dir, data_path, max_aspect_ratio)``, ``shuffle()``, ``size()``,
``nextBatch(batch_size)`` returning ``{images, targets, targets_eval, num_nonzeros, img_paths}`` -- so a reference-style
training loop reads the same.  The image arithmetic (255*rgb2y, image.scale to 32 x imgW) runs on the GPU through
``aocr_preprocess_lines``; the host keeps what the reference keeps on the host: the list file, the width buckets, the
label -> id conversion (utils.lua:104-118) and the target assembly.  There is no CPU fallback for the image path.

Differences from the reference, all deliberate:
  * image decoding (data_gen.lua:67 ``image.load``): ``.npy`` (H,W or H,W,3 uint8), binary ``.pgm`` (P5) / ``.ppm`` (P6) are read directly; every other
    file (JPEG, PNG, ...) goes through Pillow when it is installed (round 6; torch/image links libjpeg / libpng itself: PNG is lossless, JPEG pixels may
    differ by an LSB between decoder versions -- unpinned like the rest of the data path); a custom ``loader`` may be passed;
  * ``force_width=100`` reproduces data_gen.lua:78 (the reference overrides the aspect-ratio width with 100);
    ``force_width=None`` applies the aspect-ratio rule of lines 72-77;
  * ``shuffle`` uses numpy's generator instead of Lua's ``math.random`` (not reproducible across the two anyway);
  * the reference caches the scaled image per line after its first visit; here the decoded uint8 image is cached on the host
    and scaled on the device when its batch is emitted (the scaled tensor goes straight into HBM).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from ._lib import check, lib, ptr

IMG_H = 32
MIN_ASPECT = 0.5


class ImageDesc(C.Structure):
    """mirror of `aocr_image_desc` (include/aocr.h)."""
    _fields_ = [("offset", C.c_int64), ("height", C.c_int32), ("width", C.c_int32), ("channels", C.c_int32), ("reserved", C.c_int32)]


def str2numlist(label: str):
    """utils.lua:104-118."""
    out = [2]
    for ch in label.encode("latin-1"):
        out.append(ch - 97 + 13 + 1 if ch > 96 else ch - 48 + 3 + 1)
    out.append(3)
    return out


def _read_pnm(path):
    with open(path, "rb") as f:
        data = f.read()
    tok, pos = [], 0
    while len(tok) < 4:                                   # magic, width, height, maxval (comments allowed)
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(data[pos:end]); pos = end
    pos += 1
    magic, w, h, maxval = tok[0], int(tok[1]), int(tok[2]), int(tok[3])
    if magic not in (b"P5", b"P6") or maxval != 255:
        raise ValueError(f"{path}: only binary 8-bit PGM/PPM is supported")
    c = 1 if magic == b"P5" else 3
    a = np.frombuffer(data, np.uint8, count=h * w * c, offset=pos)
    return a.reshape(h, w) if c == 1 else a.reshape(h, w, 3)


def _read_pillow(path):
    """JPEG / PNG / ... through Pillow: gray files stay one channel, everything else becomes RGB (what 255 * image.rgb2y then reduces, data_gen.lua:68)."""
    from PIL import Image                                   # optional dependency: a missing Pillow makes the line unreadable (skipped, as a failed image.load is)
    with Image.open(path) as im:
        im.load()
        if im.mode in ("1", "L", "I;16", "I", "F"):
            im = im.convert("L")
        elif im.mode != "RGB":
            im = im.convert("RGB")                          # palette, RGBA (alpha dropped), CMYK, LA
        return np.asarray(im, dtype=np.uint8)


def load_image(path):
    """uint8 (H,W) or (H,W,3) array, or None when the file cannot be read (data_gen.lua:66: a failed load skips the line)."""
    try:
        if path.endswith(".npy"):
            a = np.load(path)
        elif path.lower().endswith((".pgm", ".ppm", ".pnm")):
            a = _read_pnm(path)
        else:
            a = _read_pillow(path)
        if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] != 3):
            return None
        return np.ascontiguousarray(a)
    except Exception:
        return None


def preprocess_batch(images_u8, out_w, device=None, stream=None):
    """255*rgb2y + image.scale(., out_w, 32) of a list of uint8 images on the GPU -> (n,1,32,out_w) fp32 device tensor."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    n = len(images_u8)
    desc = (ImageDesc * n)()
    off = 0
    for i, a in enumerate(images_u8):
        h, w = a.shape[0], a.shape[1]
        c = 1 if a.ndim == 2 else 3
        desc[i] = ImageDesc(off, h, w, c, 0)
        off += (h * w * c + 15) // 16 * 16
    buf = np.zeros(max(off, 16), np.uint8)
    for i, a in enumerate(images_u8):
        buf[desc[i].offset:desc[i].offset + a.size] = a.reshape(-1)
    src = torch.from_numpy(buf).to(device)
    dsc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(device)
    out = torch.empty((n, 1, IMG_H, out_w), dtype=torch.float32, device=device)
    st = stream if stream is not None else torch.cuda.current_stream(device).cuda_stream
    check(lib.aocr_preprocess_lines(st, ptr(src), ptr(dsc), n, IMG_H, out_w, ptr(out)), "aocr_preprocess_lines")
    return out


class DataGen:
    def __init__(self, data_base_dir, data_path, max_aspect_ratio, force_width=100, loader=None, device=None):
        self.imgH = IMG_H
        self.data_base_dir = data_base_dir
        self.data_path = data_path
        self.max_aspect_ratio = max_aspect_ratio
        self.min_aspect_ratio = MIN_ASPECT
        self.force_width = force_width
        self.loader = loader or load_image
        self.device = device
        path = data_path if os.path.exists(data_path) else os.path.join(data_base_dir, data_path)
        if not os.path.exists(path):
            raise FileNotFoundError(f"Error: Data file {data_path} not found ")          # data_gen.lua:33-35 (the reference exits)
        self.lines = []
        with open(path) as f:
            for line in f:
                parts = line.split()
                if len(parts) >= 2:
                    self.lines.append([parts[0], parts[1], None, None])                  # filename, label, image (cached), ids
        self.cursor = 0
        self.buffer = {}

    def shuffle(self, seed=None):
        rng = np.random.default_rng(seed)
        counter = len(self.lines)
        while counter > 1:                                                               # utils.lua:13-20
            index = int(rng.integers(1, counter + 1))
            self.lines[index - 1], self.lines[counter - 1] = self.lines[counter - 1], self.lines[index - 1]
            counter -= 1

    def size(self):
        return len(self.lines)

    def _width(self, h, w):
        aspect = min(w / h, self.max_aspect_ratio)
        aspect = max(aspect, self.min_aspect_ratio)
        img_w = int(math.ceil(aspect * self.imgH))
        return self.force_width if self.force_width is not None else img_w

    def _emit(self, img_w):
        items = self.buffer.pop(img_w)
        images = preprocess_batch([it[0] for it in items], img_w, self.device)
        max_len = max(len(it[1]) for it in items)
        targets = np.ones((len(items), max_len - 1), np.int32)
        targets_eval = np.ones((len(items), max_len - 1), np.int32)
        nnz = 0
        for i, it in enumerate(items):
            ids = it[1]
            nnz += len(ids) - 1
            targets[i, :len(ids) - 1] = ids[:-1]                                         # SOS, ch1, ..., chn
            targets_eval[i, :len(ids) - 1] = ids[1:]                                     # ch1, ..., chn, EOS
        return [images, targets, targets_eval, nnz, [it[2] for it in items]]

    def nextBatch(self, batch_size):
        while self.cursor < len(self.lines):
            ln = self.lines[self.cursor]
            if ln[2] is None:
                img = self.loader(os.path.join(self.data_base_dir, ln[0]))
                if img is not None:
                    ln[2] = img; ln[3] = str2numlist(ln[1])
            self.cursor += 1
            if ln[2] is None:
                continue
            img_w = self._width(ln[2].shape[0], ln[2].shape[1])
            self.buffer.setdefault(img_w, []).append((ln[2], ln[3], ln[0]))
            if len(self.buffer[img_w]) == batch_size:
                return self._emit(img_w)
        if not self.buffer:
            self.cursor = 0
            return None
        return self._emit(next(iter(self.buffer)))
