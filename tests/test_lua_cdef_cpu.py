"""CPU: the LuaJIT `ffi.cdef` of lua/aocr_ffi.lua against include/aocr.h, declaration by declaration.  The Lua side of the
boundary cannot be executed in this image (no Lua toolchain: SURVEY.md 8(c)); this test is what keeps it from rotting -- every
prototype, struct layout and callback typedef the cdef declares must be the header's, and every header entry point must be declared."""
import os
import re

import cdecl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _both():
    hdr = open(os.path.join(ROOT, "include", "aocr.h")).read()
    lua = open(os.path.join(ROOT, "lua", "aocr_ffi.lua")).read()
    h = cdecl.parse(hdr)
    l = cdecl.parse(cdecl.lua_cdef_blocks(lua), macros=cdecl.defines(hdr))
    return h, l


def test_parser_sees_the_whole_header():
    h, _ = _both()
    assert len(h["functions"]) >= 47 and "aocr_train_forward_backward" in h["functions"]
    assert h["functions"]["aocr_param_counts"] == ("int", ["const aocr_config*", "int64_t[5]"])
    assert h["functions"]["aocr_last_error"] == ("const char*", ["void"])
    assert h["fnptrs"]["aocr_allreduce_fn"] == ("int", ["void*", "void*", "int64_t", "int32_t", "void*"])
    assert [f for _, f in h["structs"]["aocr_config"]][:3] == ["batch_size", "img_h", "max_img_w"]


def test_every_header_entry_point_is_declared_in_the_cdef_with_the_same_signature():
    h, l = _both()
    missing = sorted(set(h["functions"]) - set(l["functions"]))
    assert not missing, f"lua/aocr_ffi.lua does not declare: {missing}"
    wrong = {n: (l["functions"][n], h["functions"][n]) for n in h["functions"] if l["functions"][n] != h["functions"][n]}
    assert not wrong, wrong
    extra = sorted(n for n in l["functions"] if n.startswith("aocr_") and n not in h["functions"])
    assert not extra, f"declared in the cdef but not in include/aocr.h: {extra}"


def test_struct_layouts_and_callback_types_agree():
    h, l = _both()
    for name, fields in h["structs"].items():
        assert l["structs"].get(name) == fields, (name, l["structs"].get(name), fields)
    for name, sig in h["fnptrs"].items():
        assert l["fnptrs"].get(name) == sig, (name, l["fnptrs"].get(name), sig)


def test_library_exports_what_the_cdef_declares():
    """and the built library exports every aocr_* symbol the cdef binds (ffi.load would fail lazily, at first use, otherwise)."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "torch-attention-ocr_amd", "aocr", "libaocr.so"))
    _, l = _both()
    for n in l["functions"]:
        if n.startswith("aocr_"):
            getattr(lib, n)


def test_lua_sources_only_call_declared_entry_points():
    _, l = _both()
    used = set()
    for f in os.listdir(os.path.join(ROOT, "lua")):
        if f.endswith(".lua"):
            used |= set(re.findall(r"\b(?:lib|L|C)\.(aocr_\w+)", open(os.path.join(ROOT, "lua", f)).read()))
    assert used and not (used - set(l["functions"])), sorted(used - set(l["functions"]))


def test_lua_files_are_block_balanced():
    """No Lua interpreter exists here; the least a test can do for the unexecuted files is a lexical check: every block opener
    (function / if / do / repeat) has its end / until, and brackets balance, after comments and string literals are stripped."""
    for f in sorted(os.listdir(os.path.join(ROOT, "lua"))):
        if not f.endswith(".lua"):
            continue
        s = open(os.path.join(ROOT, "lua", f)).read()
        s = re.sub(r"--\[\[.*?\]\]", "", s, flags=re.S)
        s = re.sub(r"ffi\.cdef\s*\[\[.*?\]\]", "", s, flags=re.S)
        s = re.sub(r"--[^\n]*", "", s)
        s = re.sub(r"'(?:\\.|[^'\\\n])*'", "''", s)
        s = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', s)
        toks = re.findall(r"\b(function|if|do|repeat|end|until)\b", s)
        opens = sum(t in ("function", "if", "do", "repeat") for t in toks)
        closes = sum(t in ("end", "until") for t in toks)
        assert opens == closes, (f, opens, closes)
        for a, b in ("()", "{}", "[]"):
            assert s.count(a) == s.count(b), (f, a, s.count(a), s.count(b))
