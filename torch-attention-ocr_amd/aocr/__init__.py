"""aocr -- host side of the MI355X-native attention-OCR hot path.

`aocr.Model` mirrors the reference's Lua `Model` class (src/model/model.lua) on top of
the C ABI in include/aocr.h (libaocr.so, hand-written HIP for gfx950).  Importing this
package loads the shared library and fails loudly if it is missing.
"""
from . import _lib                      # noqa: F401  (raises ImportError when libaocr.so is absent)
from ._lib import AocrError, Config, COMPUTE_BF16, COMPUTE_F32, lib, last_error, check, ptr, param_table
from .model import Model, eval_word_err_rate, numlist2str, GROUPS
from . import synth
from . import data
from .data import DataGen
from . import dictionary
from . import t7, checkpoint
from .dictionary import Trie, load_dictionary, build_trie, levenshtein

__all__ = ["Model", "DataGen", "data", "dictionary", "t7", "checkpoint", "Trie", "load_dictionary", "build_trie", "levenshtein", "AocrError", "Config", "COMPUTE_F32", "COMPUTE_BF16", "lib", "last_error", "check", "ptr",
           "param_table", "eval_word_err_rate", "numlist2str", "GROUPS", "synth"]
