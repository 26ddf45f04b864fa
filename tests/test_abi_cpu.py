"""CPU: the C-ABI library loads, exports every symbol include/aocr.h declares, reports the parameter layout the
oracle expects, and fails loudly (non-zero status + message) on bad arguments.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    import aocr
    hdr = open(os.path.join(ROOT, "include", "aocr.h")).read()
    names = sorted(set(re.findall(r"^(?:int|size_t|int64_t|const char\*)\s+(aocr_[a-z0-9_]+)\s*\(", hdr, re.M)))
    assert len(names) >= 28
    raw = C.CDLL(aocr._lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/aocr.h but not exported by libaocr.so"
        assert n in aocr._lib.SIGNATURES, f"{n} has no ctypes signature in aocr/_lib.py"
    assert aocr.lib.aocr_version() == 1


def test_param_layout_matches_reference_counts():
    import aocr
    import oracle_torch as O
    cfg = aocr.Config(64, 32, 100, 256, 1, 2, 39, 20, 1, 50, 5, 0)
    tab, counts = aocr.param_table(cfg)
    assert counts == [5551360, 788480, 788480, 5030668, 20007]          # CNN / enc / dec / projector
    assert sum(counts) == 12178995                                       # gradient all-reduce payload, SURVEY.md 5
    spec = O.param_spec(O.OcrConfig())
    assert [s[0] for s in spec] == [t[0] for t in tab]
    off = 0
    for (n, shape, _, _), (n2, g, o, shp) in zip(spec, tab):
        if len(shape) == 4:
            shape = (shape[0], shape[2], shape[3], shape[1])             # taps channels-last
        assert tuple(shape) == tuple(shp) and o == off and o % 4 == 0, n
        assert g == O.group_of(n)
        off += int(np.prod(shp))
    # reference defaults (He=512): 29.8 M parameters
    cfg2 = aocr.Config(400, 32, 100, 512, 1, 2, 39, 20, 1, 50, 5, 0)
    assert sum(aocr.param_table(cfg2)[1]) == 29815859


def test_workspace_and_errors():
    import aocr
    cfg = aocr.Config(64, 32, 100, 256, 1, 2, 39, 20, 1, 50, 5, 0)
    ws = aocr.lib.aocr_workspace_bytes(C.byref(cfg))
    assert 50e6 < ws < 2e9
    bad = aocr.Config(64, 32, 100, 250, 1, 2, 39, 20, 1, 50, 5, 0)       # enc_hidden not a multiple of 16
    assert aocr.lib.aocr_workspace_bytes(C.byref(bad)) == 0
    assert "enc_hidden" in aocr.last_error()
    counts = (C.c_int64 * 5)()
    assert aocr.lib.aocr_param_counts(C.byref(bad), counts) != 0
    with pytest.raises(aocr.AocrError):
        aocr.check(aocr.lib.aocr_model_create(C.byref(cfg), None, None, None, None, 0, None, C.byref(C.c_void_p())), "create")
    assert "NULL" in aocr.last_error()
    assert aocr.lib.aocr_bn_state_count() == 2 * (256 + 512 + 512)


def test_host_side_needs_gpu_loudly():
    import torch
    import aocr
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        aocr.Model().create(dict(encoder_num_hidden=32, batch_size=2))
