"""ctypes binding of libaocr.so (the C ABI declared in include/aocr.h).

There is NO fallback: if the HIP library is missing or a symbol is absent this
module raises at import time, and every entry point raises ``AocrError`` on a
non-zero status.  Nothing here touches ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- BEFORE libaocr.so: the process must end up with ONE HIP runtime, the one PyTorch loads.  Loading
#                libaocr.so first pulls in /opt/rocm's libamdhip64 next to PyTorch's own copy, and the library then sees no device

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AOCR_LIB") or os.path.join(_HERE, "libaocr.so")     # AOCR_LIB: another build of the SAME library (debug stamps), as in lua/aocr_ffi.lua

NUM_GROUPS = 5
COMPUTE_F32, COMPUTE_BF16 = 0, 1
PROF_FAMILIES = ["other", "conv_fwd", "conv_dgrad", "conv_wgrad", "bn", "pool_conv1", "encoder_seq", "rnn_gemm", "decoder_fwd",
                 "decoder_bwd", "sgd", "decode_chain"]          # AOCR_PROF_* of include/aocr.h


class AocrError(RuntimeError):
    pass


class TrieDesc(C.Structure):
    """mirror of `aocr_trie` (include/aocr.h)."""
    _fields_ = [("child_mask_dev", C.c_void_p), ("child_base_dev", C.c_void_p), ("child_dev", C.c_void_p),
                ("n_nodes", C.c_int32), ("n_edges", C.c_int32)]


class Config(C.Structure):
    """mirror of `aocr_config` (include/aocr.h)."""
    _fields_ = [(n, C.c_int32) for n in (
        "batch_size", "img_h", "max_img_w", "enc_hidden", "enc_layers", "dec_layers", "vocab", "emb",
        "input_feed", "max_decoder_l", "max_beam", "compute")]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `make -C torch-attention-ocr_amd/csrc` "
        "(or __graft_entry__.build()); there is no CPU fallback for the hot path")

lib = C.CDLL(LIB_PATH)

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_cfgp = C.POINTER(Config)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, _i64, _i32, _vp)     # aocr_allreduce_fn of include/aocr.h

# name -> (restype, argtypes); every symbol declared in include/aocr.h
SIGNATURES = {
    "aocr_last_error": (C.c_char_p, []),
    "aocr_version": (C.c_int, []),
    "aocr_param_counts": (C.c_int, [_cfgp, C.POINTER(_i64)]),
    "aocr_param_entry": (C.c_int, [_cfgp, _i32, C.c_char_p, C.POINTER(_i32), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i64)]),
    "aocr_bn_state_count": (_i64, []),
    "aocr_workspace_bytes": (_sz, [_cfgp]),
    "aocr_model_create": (C.c_int, [_cfgp, _vp, _vp, _vp, _vp, _sz, _vp, C.POINTER(_vp)]),
    "aocr_model_destroy": (C.c_int, [_vp]),
    "aocr_model_set_stream": (C.c_int, [_vp, _vp]),
    "aocr_train_forward_backward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "aocr_set_dropout": (C.c_int, [_vp, C.c_double, C.c_uint64, C.c_uint64]),
    "aocr_cluster_status": (C.c_int, [_vp, C.POINTER(_i32)]),
    "aocr_grad_buckets": (C.c_int, [_cfgp, C.POINTER(_i64), C.POINTER(_i64)]),
    "aocr_stream_wait_grads": (C.c_int, [_vp, _i32, _vp]),
    "aocr_comm_unique_id": (C.c_int, [C.c_char_p]),
    "aocr_comm_init_rank": (C.c_int, [_vp, C.c_char_p, _i32, _i32, _i32]),
    "aocr_comm_set_callback": (C.c_int, [_vp, _vp, _vp, _i32, _i32]),
    "aocr_allreduce_grads": (C.c_int, [_vp, _vp]),
    "aocr_comm_destroy": (C.c_int, [_vp]),
    "aocr_comm_exposed_ms": (C.c_int, [_vp, C.POINTER(_f32)]),
    "aocr_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "aocr_sgd_step": (C.c_int, [_vp, _f32, _f32, _vp]),
    "aocr_adadelta_step": (C.c_int, [_vp, _f32, _f32, _f32, _vp]),
    "aocr_forward_logits": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "aocr_decode": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "aocr_decode_dict": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "aocr_get_tensor": (C.c_int, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i32), C.POINTER(_i64)]),
    "aocr_profile_kernel": (C.c_int, [_vp, _i32, _i32, C.POINTER(_f32), C.POINTER(C.c_double)]),
    "aocr_profile_enable": (C.c_int, [_vp, _i32]),
    "aocr_profile_read": (C.c_int, [_vp, C.POINTER(_f32), C.POINTER(_i32)]),
    "aocr_gemm": (C.c_int, [_vp, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _i32]),
    "aocr_conv2d_forward": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp] + [_i32] * 9),
    "aocr_conv2d_backward_data": (C.c_int, [_vp, _i32, _vp, _vp, _vp] + [_i32] * 7),
    "aocr_conv2d_backward_filter": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp] + [_i32] * 7),
    "aocr_unpool_relu_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp] + [_i32] * 5),
    "aocr_conv1_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32]),
    "aocr_conv1_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32]),
    "aocr_batchnorm_relu_forward": (C.c_int, [_vp] * 9 + [_i64, _i32, _i32, _i32, _i32]),
    "aocr_batchnorm_relu_backward": (C.c_int, [_vp] * 10 + [_i64, _i32, _i32]),
    "aocr_lstm_cell_forward": (C.c_int, [_vp, _i32, _vp, _i32] + [_vp] * 9 + [_i32, _i32]),
    "aocr_lstm_cell_forward_zx": (C.c_int, [_vp, _i32, _vp, _i64] + [_vp] * 6 + [_i32, _i32]),
    "aocr_lstm_cell_backward": (C.c_int, [_vp] * 8 + [_i32, _i32]),
    "aocr_attention_forward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32]),
    "aocr_attention_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _i32]),
    "aocr_pointwise": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i64]),
    "aocr_lookup_forward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32]),
    "aocr_lookup_backward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32]),
    "aocr_logsoftmax_nll": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _f32]),
    "aocr_beam_select_dict": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "aocr_edit_distance": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "aocr_preprocess_lines": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "aocr_beam_select": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = library/header mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args


def last_error() -> str:
    return (lib.aocr_last_error() or b"").decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise AocrError(f"{what}: {last_error()}" if what else last_error())


def ptr(t):
    """device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def param_table(cfg: Config):
    """[(name, group, offset, shape)] + per-group counts, straight from the library."""
    counts = (_i64 * NUM_GROUPS)()
    check(lib.aocr_param_counts(C.byref(cfg), counts), "aocr_param_counts")
    out = []
    i = 0
    while True:
        name = C.create_string_buffer(64)
        g, nd, off = _i32(), _i32(), _i64()
        shape = (_i64 * 4)()
        rc = lib.aocr_param_entry(C.byref(cfg), i, name, C.byref(g), C.byref(off), C.byref(nd), shape)
        if rc == 1:
            break
        if rc != 0:
            raise AocrError(last_error())
        out.append((name.value.decode(), g.value, off.value, tuple(shape[k] for k in range(nd.value))))
        i += 1
    return out, [counts[k] for k in range(NUM_GROUPS)]
