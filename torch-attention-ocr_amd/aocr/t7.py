"""aocr.t7 -- Torch7 binary serialization (`torch.save` / `torch.load`, the wire format of the reference's checkpoints,
src/model/model.lua:51,724), reader and writer, host side only.

The format lives in the un-vendored torch7 package ([upstream] `File.lua` writeObject/readObject, `generic/Tensor.c` and
`generic/Storage.c` write), version un-pinned by the reference; restated here from its published layout (little-endian,
native 8-byte `long`):

    object   := int32 type, payload
    type 0   nil
    type 1   number      float64
    type 2   string      int32 length, bytes
    type 3   table       int32 index [, int32 n, n x (object key, object value)]        -- body only the first time an index is seen
    type 4   torch obj   int32 index [, string "V <n>", string class name, class payload]
    type 5   boolean     int32 0/1
    type 6,7,8 function  int32 index [, int32 length, bytes of string.dump, object upvalues]
    class payload        Tensor:  int32 ndim, int64 size[ndim], int64 stride[ndim], int64 storageOffset (1-based), object storage
                         Storage: int64 n, n raw elements
                         any other class (nn.*, cudnn.*, nngraph.*, graph.*): one object = the table of its fields

Objects sharing an index are the same object (aliasing and cycles survive a round trip).  Lua values map to Python as:
nil None, number float (int when integral), string str (latin-1), boolean bool, table `LuaTable` (a dict; `.list()` for 1..n keys), tensors numpy
arrays (views of a shared storage array), other torch objects `TorchObject(typename, fields)`, functions `LuaFunction`.
"""
from __future__ import annotations

import struct
from typing import Any, Dict

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN, TYPE_FUNCTION, TYPE_LEGACY_RECUR, TYPE_RECUR = range(9)

_ELEM = {"Byte": np.uint8, "Char": np.int8, "Short": np.int16, "Int": np.int32, "Long": np.int64, "Float": np.float32,
         "Double": np.float64, "Half": np.float16}


def _elem_of(typename: str, suffix: str):
    """numpy dtype of 'torch.<Kind><suffix>' (Cuda kinds share the host layout: torch.CudaTensor is float32) or None."""
    if not (typename.startswith("torch.") and typename.endswith(suffix)):
        return None
    kind = typename[len("torch."):-len(suffix)]
    if kind.startswith("Cuda"):
        kind = kind[4:] or "Float"
    return _ELEM.get(kind)


class LuaTable(dict):
    """a Lua table; integer keys are Python ints (Lua numbers with integral value)."""

    def list(self):
        n = len(self)
        if any(k not in self for k in range(1, n + 1)):
            raise ValueError("table is not a 1..n array")
        return [self[k] for k in range(1, n + 1)]

    def array_part(self):
        out, i = [], 1
        while i in self:
            out.append(self[i]); i += 1
        return out

    __hash__ = object.__hash__            # tables are keys of other tables by identity (nngraph's mapindex)
    __eq__ = object.__eq__
    __ne__ = object.__ne__


class TorchObject:
    def __init__(self, typename: str, fields: Any = None, version: int = 1):
        self.typename, self.fields, self.version = typename, fields if fields is not None else LuaTable(), version

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default) if isinstance(self.fields, dict) else default

    def __repr__(self):
        return f"<{self.typename}>"


class Storage:
    """A bare torch.*Storage (host side: never a Cuda type) -- e.g. nn.View's `size` field is a torch.LongStorage, not a tensor."""

    def __init__(self, array):
        self.array = np.ascontiguousarray(np.asarray(array).reshape(-1))


class LuaFunction:
    def __init__(self, dumped: bytes, upvalues: Any, kind: int = TYPE_RECUR):
        self.dumped, self.upvalues, self.kind = dumped, upvalues, kind


class T7Error(ValueError):
    pass


class Reader:
    def __init__(self, data: bytes):
        self.b, self.o, self.objects = memoryview(data), 0, {}

    def _take(self, n):
        if self.o + n > len(self.b):
            raise T7Error(f"truncated file: need {n} bytes at offset {self.o}")
        v = self.b[self.o:self.o + n]; self.o += n
        return v

    def _int(self):
        return struct.unpack("<i", self._take(4))[0]

    def _long(self):
        return struct.unpack("<q", self._take(8))[0]

    def _str(self):
        return bytes(self._take(self._int())).decode("latin-1")

    def read(self):
        t = self._int()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            v = struct.unpack("<d", self._take(8))[0]
            return int(v) if v == v and abs(v) < 2 ** 53 and v == int(v) else v      # integral Lua numbers become ints (table keys 1..n)
        if t == TYPE_STRING:
            return self._str()
        if t == TYPE_BOOLEAN:
            return self._int() != 0
        if t in (TYPE_TABLE, TYPE_TORCH, TYPE_FUNCTION, TYPE_LEGACY_RECUR, TYPE_RECUR):
            idx = self._int()
            if idx in self.objects:
                return self.objects[idx]
            if t == TYPE_TABLE:
                tab = LuaTable(); self.objects[idx] = tab
                for _ in range(self._int()):
                    k = self.read(); tab[k] = self.read()
                return tab
            if t == TYPE_TORCH:
                return self._torch(idx)
            n = self._int(); dumped = bytes(self._take(n))
            fn = LuaFunction(dumped, None, t); self.objects[idx] = fn
            fn.upvalues = self.read()
            return fn
        raise T7Error(f"unknown type tag {t} at offset {self.o - 4}")

    def _torch(self, idx):
        version = self._str()
        if version.startswith("V "):
            vnum = int(version[2:]); cls = self._str()
        else:                                            # legacy files carry no version string
            vnum, cls = 0, version
        dt = _elem_of(cls, "Tensor")
        if dt is not None:
            nd = self._int()
            size = [self._long() for _ in range(nd)]; stride = [self._long() for _ in range(nd)]
            off = self._long() - 1
            holder = [None]; self.objects[idx] = holder     # a tensor cannot refer to itself; patched below
            st = self.read()
            if st is None or nd == 0:
                arr = np.zeros((0,), dt)
            else:
                need = off + sum((s - 1) * k for s, k in zip(size, stride)) + 1 if all(s > 0 for s in size) else 0
                if need > st.shape[0] or off < 0 or any(k < 0 for k in stride):
                    raise T7Error(f"{cls}: view (size {size}, stride {stride}, offset {off}) exceeds its storage of {st.shape[0]}")
                arr = np.lib.stride_tricks.as_strided(st[off:], shape=size, strides=[k * st.itemsize for k in stride], writeable=False)
            self.objects[idx] = arr
            return arr
        dt = _elem_of(cls, "Storage")
        if dt is not None:
            n = self._long()
            arr = np.frombuffer(self._take(n * np.dtype(dt).itemsize), dtype=dt)
            self.objects[idx] = arr
            return arr
        obj = TorchObject(cls, None, vnum); self.objects[idx] = obj
        obj.fields = self.read()
        return obj


class Writer:
    """Python -> Torch7.  dict / LuaTable -> table, list / tuple -> 1..n table, numpy array -> tensor of the matching kind
    (`cuda=True`: float32 arrays become torch.CudaTensor like the reference's nets), TorchObject -> its class."""

    def __init__(self, cuda: bool = False):
        self.out, self.index, self.n, self.cuda, self.keep = bytearray(), {}, 0, cuda, []

    def _int(self, v):
        self.out += struct.pack("<i", v)

    def _long(self, v):
        self.out += struct.pack("<q", v)

    def _str(self, s):
        b = s.encode("latin-1") if isinstance(s, str) else bytes(s)
        self._int(len(b)); self.out += b

    def _ref(self, obj, tag):
        """writes the tag and the index; True if the body still has to follow."""
        self._int(tag)
        key = id(obj)
        if key in self.index:
            self._int(self.index[key]); return False
        self.n += 1; self.index[key] = self.n; self.keep.append(obj)
        self._int(self.n); return True

    def _tensor_name(self, dt, suffix):
        for kind, d in _ELEM.items():
            if np.dtype(d) == dt:
                if self.cuda:
                    return "torch.Cuda" + ("" if kind == "Float" else kind) + suffix
                return f"torch.{kind}{suffix}"
        raise T7Error(f"no Torch7 tensor type for dtype {dt}")

    def write(self, obj):
        if obj is None:
            self._int(TYPE_NIL)
        elif isinstance(obj, (bool, np.bool_)):
            self._int(TYPE_BOOLEAN); self._int(1 if obj else 0)
        elif isinstance(obj, (int, float, np.integer, np.floating)):
            self._int(TYPE_NUMBER); self.out += struct.pack("<d", float(obj))
        elif isinstance(obj, (str, bytes)):
            self._int(TYPE_STRING); self._str(obj)
        elif isinstance(obj, np.ndarray):
            if self._ref(obj, TYPE_TORCH):
                self._str("V 1"); self._str(self._tensor_name(obj.dtype, "Tensor"))
                a = np.ascontiguousarray(obj)
                self._int(a.ndim)
                for s in a.shape:
                    self._long(s)
                for k in a.strides:
                    self._long(k // a.itemsize)
                self._long(1)
                if a.ndim == 0 or a.size == 0:
                    self._int(TYPE_NIL) if a.ndim == 0 else self._storage(a.reshape(-1))
                else:
                    self._storage(a.reshape(-1))
        elif isinstance(obj, Storage):
            if self._ref(obj, TYPE_TORCH):
                cuda, self.cuda = self.cuda, False
                self._str("V 1"); self._str(self._tensor_name(obj.array.dtype, "Storage"))
                self.cuda = cuda
                self._long(obj.array.shape[0]); self.out += obj.array.tobytes()
        elif isinstance(obj, TorchObject):
            if self._ref(obj, TYPE_TORCH):
                self._str(f"V {obj.version}"); self._str(obj.typename)
                self.write(obj.fields)
        elif isinstance(obj, LuaFunction):
            if self._ref(obj, obj.kind):
                self._int(len(obj.dumped)); self.out += obj.dumped
                self.write(obj.upvalues)
        elif isinstance(obj, dict):
            if self._ref(obj, TYPE_TABLE):
                self._int(len(obj))
                for k, v in obj.items():
                    self.write(k); self.write(v)
        elif isinstance(obj, (list, tuple)):
            if self._ref(obj, TYPE_TABLE):
                self._int(len(obj))
                for i, v in enumerate(obj, 1):
                    self.write(i); self.write(v)
        else:
            raise T7Error(f"cannot serialize {type(obj).__name__}")

    def _storage(self, flat):
        st = np.ascontiguousarray(flat); self.keep.append(st)
        self._int(TYPE_TORCH); self.n += 1; self._int(self.n)
        self._str("V 1"); self._str(self._tensor_name(st.dtype, "Storage"))
        self._long(st.shape[0]); self.out += st.tobytes()


def loads(data: bytes):
    r = Reader(data)
    obj = r.read()
    return obj


def load(path: str):
    with open(path, "rb") as f:
        return loads(f.read())


def dumps(obj, cuda: bool = False) -> bytes:
    w = Writer(cuda); w.write(obj)
    return bytes(w.out)


def save(path: str, obj, cuda: bool = False):
    with open(path, "wb") as f:
        f.write(dumps(obj, cuda))
