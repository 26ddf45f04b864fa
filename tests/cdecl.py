"""A small C-declaration normaliser (test infrastructure): turns the prototypes, structs and function-pointer typedefs of a header
or of a LuaJIT `ffi.cdef[[ ... ]]` block into comparable signatures, so that the unexecuted Lua side of the boundary
(lua/aocr_ffi.lua) cannot drift from include/aocr.h unnoticed."""
import re


def _strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return text


def defines(text):
    out = {}
    for m in re.finditer(r"^\s*#\s*define\s+(\w+)\s+(-?\d+)\s*$", _strip(text), flags=re.M):
        out[m.group(1)] = m.group(2)
    return out


def _norm_type(tokens):
    return " ".join(tokens).replace(" *", "*").replace("* ", "*")


def _param(p, macros):
    """'const float* x_dev' -> 'const float*'; 'int64_t shape[4]' -> 'int64_t[4]'; 'void' -> 'void'."""
    p = p.strip()
    dims = re.findall(r"\[\s*(\w*)\s*\]", p)
    p = re.sub(r"\[\s*\w*\s*\]", "", p).strip()
    toks = re.findall(r"\w+|\*", p)
    if len(toks) > 1 and toks[-1] != "*" and toks[-1] not in ("int", "char", "float", "double", "void", "size_t") and not toks[-1].endswith("_t"):
        toks = toks[:-1]                                         # drop the parameter name
    t = _norm_type(toks)
    for d in dims:
        t += "[%s]" % macros.get(d, d)
    return t


def parse(text, macros=None):
    """{"functions": {name: (ret, [param types])}, "structs": {name: [(type, field)]}, "fnptrs": {name: (ret, [params])}}"""
    macros = dict(macros or {}); macros.update(defines(text))
    src = _strip(text)
    src = re.sub(r"^\s*#[^\n]*$", " ", src, flags=re.M)
    src = re.sub(r'extern\s+"C"\s*\{', " ", src)
    out = {"functions": {}, "structs": {}, "fnptrs": {}}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = [n.strip() for n in decl.split(",")]
            first = re.findall(r"\w+|\*", names[0])
            base, fname = first[:-1], first[-1]
            fields.append((_norm_type(base), fname))
            for n in names[1:]:
                fields.append((_norm_type(base), n))
        out["structs"][m.group(3)] = fields
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    for m in re.finditer(r"typedef\s+([\w\s\*]+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", src, flags=re.S):
        out["fnptrs"][m.group(2)] = (_norm_type(re.findall(r"\w+|\*", m.group(1))), [_param(p, macros) for p in m.group(3).split(",")])
    src = re.sub(r"typedef[^;]*;", " ", src, flags=re.S)
    for m in re.finditer(r"([\w\s\*]+?)\b(\w+)\s*\(([^()]*)\)\s*;", src, flags=re.S):
        ret = _norm_type(re.findall(r"\w+|\*", m.group(1)))
        if not ret:
            continue
        out["functions"][m.group(2)] = (ret, [_param(p, macros) for p in m.group(3).split(",")])
    return out


def lua_cdef_blocks(text):
    return "\n".join(m.group(1) for m in re.finditer(r"ffi\.cdef\s*\[\[(.*?)\]\]", text, flags=re.S))
