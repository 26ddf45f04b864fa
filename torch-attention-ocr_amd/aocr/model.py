"""Host-side mirror of the reference's `Model` class (src/model/model.lua:18-731).

Same method names, argument meaning and error behaviour as the Lua class that
src/train.lua drives (`create`, `load`, `step`, `save`, `vis`, `shutdown`, fields
`global_step`, `optim_state.learningRate`), sitting directly on the C ABI of libaocr.
PyTorch is used only as plumbing: device allocations, the HIP stream and
`torch.distributed` (RCCL) for the data-parallel gradient all-reduce.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, dist
from ._lib import Config, check, lib, ptr
from .dictionary import edit_distance_device

PAD, GO, EOS = 1, 2, 3
GROUPS = ["cnn", "enc_fw", "enc_bw", "dec", "proj"]

_DEFAULTS = dict(  # src/train.lua:41-62
    dropout=0.0, encoder_num_hidden=512, encoder_num_layers=1, decoder_num_layers=2, target_vocab_size=39,
    target_embedding_size=20, max_encoder_l=80, max_decoder_l=50, input_feed=False, batch_size=400, prealloc=False,
    learning_rate=0.1, seed=910820, img_h=32, max_img_w=100, max_beam=5, compute="f32")


def _cfg_get(config, key):
    if isinstance(config, dict):
        return config.get(key, _DEFAULTS[key])
    return getattr(config, key, _DEFAULTS[key])


def numlist2str(ids):
    """src/utils/utils.lua:119-134."""
    return "".join(chr(v - 1 - 13 + 97) if v > 13 else chr(v - 1 - 3 + 48) for v in ids)


def eval_word_err_rate(labels: np.ndarray, target_labels: np.ndarray, visualize: bool = False):
    """evalWordErrRate, src/utils/utils.lua:136-175: sequences are cut at the first EOS (3) and compared exactly."""
    B = labels.shape[0]
    werr, pred, gold = 0.0, [], []
    for b in range(B):
        def cut(row):
            out = []
            for v in row:
                if v == EOS:
                    break
                out.append(int(v))
            return out
        p, g = cut(labels[b]), cut(target_labels[b])
        if visualize:
            pred.append(numlist2str(p)); gold.append(numlist2str(g))
        if p != g:
            werr += 1
    return werr, pred, gold


class Model:
    def __init__(self):
        self._h = None
        self.visualize = False
        self.visualize_file = None
        self.global_step = 0
        self.optim_state = {}
        self.last_norms = None
        self._comm_stream = None
        self._buckets = None
        self._drop_last_gs, self._drop_repeat = None, 0

    # ------------------------------------------------------------------ construction
    def _set_structure(self, config):
        g = lambda k: _cfg_get(config, k)
        self.cnn_feature_size = 512
        self.dropout = float(g("dropout"))
        if not 0.0 <= self.dropout < 1.0:
            raise ValueError(f"dropout {self.dropout} outside [0, 1)")
        self.dropout_seed = 910820          # the mask is a counter-based function of (seed, global_step, site, element): include/aocr.h
        self.encoder_num_hidden = int(g("encoder_num_hidden"))
        self.encoder_num_layers = int(g("encoder_num_layers"))
        self.decoder_num_hidden = 2 * self.encoder_num_hidden
        self.decoder_num_layers = int(g("decoder_num_layers"))
        self.target_vocab_size = int(g("target_vocab_size"))
        self.target_embedding_size = int(g("target_embedding_size"))
        self.input_feed = bool(g("input_feed"))

    def _set_runtime(self, config):
        g = lambda k: _cfg_get(config, k)
        self.max_encoder_l = int(g("max_encoder_l"))
        self.max_decoder_l = int(g("max_decoder_l"))
        self.batch_size = int(g("batch_size"))
        self.prealloc = bool(g("prealloc"))
        self.img_h = int(g("img_h"))
        self.max_img_w = int(g("max_img_w"))
        self.max_beam = int(g("max_beam"))
        comp = g("compute")
        self.compute = _lib.COMPUTE_BF16 if comp in ("bf16", 1) else _lib.COMPUTE_F32

    def create(self, config):
        """model:create, model.lua:83-112: fresh parameters (Torch7 default init distributions [upstream])."""
        self._set_structure(config)
        self._set_runtime(config)
        self.global_step = 0
        self.optim_state = {"learningRate": float(_cfg_get(config, "learning_rate"))}
        self._build()
        self._init_parameters(int(_cfg_get(config, "seed")))
        return self

    def load(self, model_path, config=None):
        """model:load, model.lua:45-80: structure from the checkpoint, max_*_l / batch_size from `config`."""
        config = config or {}
        assert os.path.isfile(model_path), f"Model {model_path} does not exist!"
        with open(model_path, "rb") as f:
            magic = f.read(4)
        if magic[:2] != b"PK":                       # not a torch.save zip: a Torch7 file (the reference's checkpoint or our flat export)
            return self._load_t7(model_path, config)
        ck = torch.load(model_path, map_location="cpu", weights_only=True)        # tensors + plain containers only: no pickle execution
        self._set_structure(ck["config"])
        merged = dict(ck["config"])
        for k in ("max_encoder_l", "max_decoder_l", "batch_size", "prealloc", "img_h", "max_img_w", "max_beam", "compute"):
            v = config.get(k) if isinstance(config, dict) else getattr(config, k, None)
            if v is not None:
                merged[k] = v
        self._set_runtime(merged)
        self.global_step = ck["global_step"]
        self.optim_state = dict(ck["optim_state"])
        self.optim_state.setdefault("learningRate", float(merged.get("learning_rate", _DEFAULTS["learning_rate"])))   # train.lua:87
        self._build()
        self.params.copy_(ck["params"].to(self.params.device))
        self.bn_state.copy_(ck["bn_state"].to(self.params.device))
        return self

    def _load_t7(self, model_path, config):
        """model:load on a Torch7-serialized checkpoint (model.lua:51-59): aocr.checkpoint pulls the tensors out of the nets."""
        from .checkpoint import read_t7_checkpoint
        ck = read_t7_checkpoint(model_path)
        merged = dict(_DEFAULTS); merged.update({k: v for k, v in ck["config"].items() if v is not None})
        self._set_structure(merged)
        for k in ("max_encoder_l", "max_decoder_l", "batch_size", "prealloc", "img_h", "max_img_w", "max_beam", "compute"):
            v = config.get(k) if isinstance(config, dict) else getattr(config, k, None)
            if v is not None:
                merged[k] = v
        self._set_runtime(merged)
        self.global_step = ck["global_step"]
        self.optim_state = dict(ck["optim_state"] or {})
        self.optim_state.setdefault("learningRate", float(merged["learning_rate"]))                                  # train.lua:87
        self._build()
        self.set_parameters({k: torch.from_numpy(v) for k, v in ck["params"].items()},
                            {k: torch.from_numpy(v) for k, v in ck["bn_state"].items()})
        return self

    def _build(self):
        """model:_build, model.lua:115-223: flat parameter/gradient vectors + workspace, no cloned cells."""
        if not torch.cuda.is_available():
            raise RuntimeError("aocr.Model needs a HIP device: the hot path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        assert self.target_vocab_size <= 40
        self.config = dict(
            dropout=self.dropout, encoder_num_hidden=self.encoder_num_hidden, encoder_num_layers=self.encoder_num_layers,
            decoder_num_hidden=self.decoder_num_hidden, decoder_num_layers=self.decoder_num_layers,
            target_vocab_size=self.target_vocab_size, target_embedding_size=self.target_embedding_size,
            max_encoder_l=self.max_encoder_l, max_decoder_l=self.max_decoder_l, input_feed=self.input_feed,
            batch_size=self.batch_size, prealloc=self.prealloc, img_h=self.img_h, max_img_w=self.max_img_w,
            max_beam=self.max_beam, compute="bf16" if self.compute == _lib.COMPUTE_BF16 else "f32")
        self.ccfg = Config(self.batch_size, self.img_h, self.max_img_w, self.encoder_num_hidden, self.encoder_num_layers,
                           self.decoder_num_layers, self.target_vocab_size, self.target_embedding_size,
                           int(self.input_feed), self.max_decoder_l, self.max_beam, self.compute)
        self.table, self.group_counts = _lib.param_table(self.ccfg)
        self.num_params = sum(self.group_counts)
        self.group_offsets = np.concatenate([[0], np.cumsum(self.group_counts)]).astype(np.int64)
        dev = self.device
        self.params = torch.zeros(self.num_params, dtype=torch.float32, device=dev)       # model.lua:163-168
        self.grad_params = torch.zeros(self.num_params, dtype=torch.float32, device=dev)
        self.bn_state = torch.zeros(lib.aocr_bn_state_count(), dtype=torch.float32, device=dev)
        self._reset_bn_state()
        ws = lib.aocr_workspace_bytes(C.byref(self.ccfg))
        if ws == 0:
            raise _lib.AocrError(_lib.last_error())
        self.workspace = torch.empty(ws, dtype=torch.uint8, device=dev)
        self._scal = torch.zeros(16, dtype=torch.float32, device=dev)                       # loss + norms
        h = C.c_void_p()
        check(lib.aocr_model_create(C.byref(self.ccfg), ptr(self.params), ptr(self.grad_params), ptr(self.bn_state),
                                    ptr(self.workspace), ws, self._stream(), C.byref(h)), "aocr_model_create")
        self._h = h
        self._dec_out = None
        self.init_beam = False

    def _reset_bn_state(self):
        st = torch.zeros_like(self.bn_state)
        o = 0
        for c in (256, 512, 512):
            st[o + c:o + 2 * c] = 1.0           # running_var = 1
            o += 2 * c
        self.bn_state.copy_(st)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _init_parameters(self, seed: int):
        gen = torch.Generator().manual_seed(seed)
        host = torch.zeros(self.num_params, dtype=torch.float32)
        for name, group, off, shape in self.table:
            n = int(np.prod(shape))
            leaf = name.split(".")[-1]
            if "conv" in name:
                cout, kh, kw, cin = self._conv_shape(name)
                stdv = 1.0 / math.sqrt(kh * kw * cin)
            elif ".bn" in name:
                stdv = None
            elif name == "dec.lookup":
                stdv = None
            elif len(shape) == 2:                                  # nn.Linear / LinearNoBias weight: 1/sqrt(fan_in)
                stdv = 1.0 / math.sqrt(shape[1])
            else:                                                  # bias: same stdv as its weight
                wshape = next(s for (nm, _, _, s) in self.table if nm == name[:-2] + ".w")
                stdv = 1.0 / math.sqrt(wshape[1])
            if name == "dec.lookup":
                v = torch.randn(n, generator=gen)
            elif ".bn" in name:
                v = torch.rand(n, generator=gen) if leaf == "w" else torch.zeros(n)
            else:
                v = (torch.rand(n, generator=gen) * 2 - 1) * stdv
            host[off:off + n] = v
        self.params.copy_(host.to(self.device))

    def _conv_shape(self, name):
        base = name[:-2]
        return next(s for (nm, _, _, s) in self.table if nm == base + ".w")

    # ------------------------------------------------------------------ parameter import / export (Torch7 layouts)
    def set_parameters(self, named: Dict[str, torch.Tensor], bn_state: Optional[Dict[str, torch.Tensor]] = None):
        """Inject weights given in Torch7 layouts (conv [Cout][Cin][kH][kW], Linear [out][in])."""
        host = torch.zeros(self.num_params, dtype=torch.float32)
        for name, group, off, shape in self.table:
            t = named[name].detach().to(torch.float32).cpu()
            if len(shape) == 4:
                t = t.permute(0, 2, 3, 1).contiguous()           # -> [Cout][kH][kW][Cin]
            assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), shape)
            host[off:off + t.numel()] = t.reshape(-1)
        self.params.copy_(host.to(self.device))
        if bn_state is not None:
            parts = []
            for i in (3, 5, 7):
                parts += [bn_state[f"cnn.bn{i}.rm"].float().reshape(-1), bn_state[f"cnn.bn{i}.rv"].float().reshape(-1)]
            self.bn_state.copy_(torch.cat(parts).to(self.device))

    def _export(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        host = flat.detach().cpu()
        out = {}
        for name, group, off, shape in self.table:
            n = int(np.prod(shape))
            t = host[off:off + n].reshape(shape)
            if len(shape) == 4:
                t = t.permute(0, 3, 1, 2).contiguous()           # -> Torch7 [Cout][Cin][kH][kW]
            out[name] = t.clone()
        return out

    def get_parameters(self):
        return self._export(self.params)

    def get_gradients(self):
        return self._export(self.grad_params)

    def get_bn_state(self):
        host = self.bn_state.detach().cpu()
        out, o = {}, 0
        for i, c in ((3, 256), (5, 512), (7, 512)):
            out[f"cnn.bn{i}.rm"] = host[o:o + c].clone(); out[f"cnn.bn{i}.rv"] = host[o + c:o + 2 * c].clone()
            o += 2 * c
        return out

    def get_tensor(self, name: str) -> torch.Tensor:
        """parity tap: copy of a named intermediate of the last step."""
        p, nd, shape = C.c_void_p(), C.c_int32(), (C.c_int64 * 4)()
        check(lib.aocr_get_tensor(self._h, name.encode(), C.byref(p), C.byref(nd), shape), "aocr_get_tensor")
        shp = tuple(shape[i] for i in range(nd.value))
        n = int(np.prod(shp))
        off = p.value - self.workspace.data_ptr()                 # every tap lives inside the workspace arena
        assert 0 <= off and off + 4 * n <= self.workspace.numel() and off % 4 == 0
        return self.workspace[off:off + 4 * n].view(torch.float32).reshape(shp).cpu()

    def get_tensor_view(self, name: str, dtype=torch.int32) -> torch.Tensor:
        """A named tap IN PLACE (device view into the workspace, 4-byte elements): tests write the cluster kernels' status word through it."""
        p, nd, shape = C.c_void_p(), C.c_int32(), (C.c_int64 * 4)()
        check(lib.aocr_get_tensor(self._h, name.encode(), C.byref(p), C.byref(nd), shape), "aocr_get_tensor")
        n = int(np.prod([shape[i] for i in range(nd.value)]))
        off = p.value - self.workspace.data_ptr()
        assert 0 <= off and off + 4 * n <= self.workspace.numel() and off % 4 == 0
        return self.workspace[off:off + 4 * n].view(dtype)

    # ------------------------------------------------------------------ one step, model.lua:226-706
    def _upload(self, batch):
        dev = self.device
        images = torch.as_tensor(batch[0]).to(device=dev, dtype=torch.float32).contiguous()     # localize(), utils.lua:96-102
        targets = torch.as_tensor(batch[1]).to(device=dev, dtype=torch.int32).contiguous()
        targets_eval = torch.as_tensor(batch[2]).to(device=dev, dtype=torch.int32).contiguous()
        assert images.dim() == 4 and images.shape[1] == 1 and images.shape[2] == self.img_h
        return images, targets, targets_eval

    def train_step_device(self, images, targets, targets_eval, global_batch=None):
        """One optimisation step on inputs already resident in HBM; enqueues only (no host sync) and returns
        the device scalar holding the step's NLL sum (summed over ranks under data parallelism).
        global_batch: rows of the step over ALL ranks (default: this rank's rows x world size, i.e. equal slices)."""
        B, _, _, W = images.shape
        target_l = targets.shape[1]
        assert target_l <= self.max_decoder_l, f"max_decoder_l ({self.max_decoder_l}) < target_l ({target_l})!"
        check(lib.aocr_model_set_stream(self._h, self._stream()))
        loss_dev = self._scal[0:1]
        if dist.world_size() > 1 and not os.environ.get("AOCR_PY_EXCHANGE"):
            dist.attach(self, sync_bn=not os.environ.get("AOCR_NO_SYNC_BN"))       # once: torch.distributed becomes the library's provider
        # d(loss) / (global batch): model.lua:645-647 divides by the step's batch size
        scale = dist.grad_scale(B) if global_batch is None else 1.0 / float(global_batch)
        self._arm_dropout()
        check(lib.aocr_train_forward_backward(self._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, target_l,
                                              scale, ptr(loss_dev)), "aocr_train_forward_backward")
        # the one exchange step of data parallelism (RCCL over xGMI), between feval and the per-group clip; bucketed in the order
        # the backward pass completes the gradient vector and run on a second stream beside it
        if getattr(self, "_comm_cb", None) is not None or getattr(self, "_comm_rccl", False):
            check(lib.aocr_allreduce_grads(self._h, ptr(loss_dev)), "aocr_allreduce_grads")   # inside the library (include/aocr.h)
        elif dist.world_size() > 1 and os.environ.get("AOCR_NO_OVERLAP"):
            dist.exchange(self.grad_params, loss_dev)             # one all-reduce after the backward pass (A/B against the overlap)
        elif dist.world_size() > 1:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.device)
                self._buckets = dist.bucket_ranges(self.ccfg)
            dist.exchange_overlapped(self.grad_params, loss_dev, self._buckets, self._wait_bucket, self._comm_stream)
        if dist.world_size() > 1 and getattr(self, "_comm_cb", None) is None and not getattr(self, "_comm_rccl", False):
            self._exchange_timeout_flag()
        norms = self._scal[2:12]
        check(lib.aocr_sgd_step(self._h, float(self.optim_state["learningRate"]), 5.0, ptr(norms)), "aocr_sgd_step")
        self.last_norms = norms
        return loss_dev

    def _exchange_timeout_flag(self):
        """Host-side exchange paths (AOCR_PY_EXCHANGE): a caller that sums the gradient buckets itself must also make the cluster
        kernels' time-out code a GLOBAL decision, as aocr_allreduce_grads does inside the library (include/aocr.h, aocr_cluster_status):
        MAX over ranks of the status word, so every rank's optimizer skips (or applies) the update together."""
        try:
            flag = self.get_tensor_view("cl_err")[0:1]
        except RuntimeError:
            return                                                # configuration without whole-sequence kernels: nothing can time out
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)

    def _wait_bucket(self, k, stream):
        check(lib.aocr_stream_wait_grads(self._h, k, C.c_void_p(stream.cuda_stream)), "aocr_stream_wait_grads")

    def decode_device(self, images, targets, targets_eval, beam_size=1, trie=None):
        """forward_only feval on device-resident inputs; enqueues only.  Returns device tensors
        (labels (B,max_decoder_l) int32, beam scores (B), gold scores (B), gold-pass NLL sum (1)).
        trie: an `aocr.dictionary.Trie` already on this device (-use_dictionary, model.lua:380-387,405-445,460-513) or None."""
        B, _, _, W = images.shape
        Lt = self.max_decoder_l
        check(lib.aocr_model_set_stream(self._h, self._stream()))
        labels = torch.empty((B, Lt), dtype=torch.int32, device=self.device)
        scores = torch.empty(B, dtype=torch.float32, device=self.device)
        gold = torch.empty(B, dtype=torch.float32, device=self.device)
        loss_dev = self._scal[0:1]
        if trie is None:
            check(lib.aocr_decode(self._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, targets.shape[1], beam_size,
                                  ptr(labels), ptr(scores), ptr(gold), ptr(loss_dev)), "aocr_decode")
        else:
            desc = trie.desc()
            check(lib.aocr_decode_dict(self._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, targets.shape[1], beam_size,
                                       C.byref(desc), ptr(labels), ptr(scores), ptr(gold), ptr(loss_dev)), "aocr_decode_dict")
        return labels, scores, gold, loss_dev

    def profile_kernel(self, which=0, iters=20):
        """HIP-event timing of one hot kernel at the last step's shape: (ms per launch, flops per launch)."""
        ms, fl = C.c_float(), C.c_double()
        check(lib.aocr_profile_kernel(self._h, which, iters, C.byref(ms), C.byref(fl)), "aocr_profile_kernel")
        return ms.value, fl.value

    def profile_families(self, fn, repeats=3):
        """Per-family HIP-event timing (include/aocr.h AOCR_PROF_*): runs fn() `repeats` times with the library's marks on and
        returns {family: ms per call}."""
        check(lib.aocr_profile_enable(self._h, 1), "aocr_profile_enable")
        try:
            tot = np.zeros(len(_lib.PROF_FAMILIES))
            for _ in range(repeats):
                fn()
                ms, marks = (C.c_float * len(_lib.PROF_FAMILIES))(), C.c_int32()
                check(lib.aocr_profile_read(self._h, ms, C.byref(marks)), "aocr_profile_read")
                tot += np.array(list(ms))
        finally:
            check(lib.aocr_profile_enable(self._h, 0), "aocr_profile_enable")
        return {k: float(v / repeats) for k, v in zip(_lib.PROF_FAMILIES, tot)}

    def step(self, batch, forward_only, beam_size=None, trie=None):
        """Returns (loss*batch_size, [num_nonzeros, num_correct]) exactly like model:step (model.lua:695-705)."""
        images, targets, targets_eval = self._upload(batch)
        if trie is not None and trie._dev is None:
            trie.to(self.device)
        num_nonzeros = batch[3]
        B, _, _, W = images.shape
        target_l = targets.shape[1]
        assert target_l <= self.max_decoder_l, f"max_decoder_l ({self.max_decoder_l}) < target_l ({target_l})!"
        check(lib.aocr_model_set_stream(self._h, self._stream()))
        loss_dev = self._scal[0:1]
        if not forward_only:
            global_batch = None
            if dist.world_size() > 1:
                # tail batches differ between ranks (data_gen.lua:123-153): the rows and label counts of the step are summed first, so
                # that d(loss) is scaled by 1 / (global rows) and train.lua:103,120 divides the summed loss by the summed count
                cnt = torch.tensor([float(B), float(num_nonzeros)], dtype=torch.float64, device=self.device)
                torch.distributed.all_reduce(cnt)
                global_batch, num_nonzeros = int(cnt[0].item()), int(cnt[1].item())
            loss_dev = self.train_step_device(images, targets, targets_eval, global_batch)
            loss = float(loss_dev.item())
            self.check_health()
            return loss, [num_nonzeros, 0.0]
        # forward only: beam search + gold pass
        beam_size = beam_size or 1
        beam_size = min(beam_size, self.target_vocab_size)
        Lt = self.max_decoder_l
        labels, scores, gold, loss_dev = self.decode_device(images, targets, targets_eval, beam_size, trie)
        # word scoring on device (evalWordErrRate, utils.lua:136-175): one Levenshtein distance per row, a word is right iff it is 0
        tge_dev = torch.full((B, Lt), PAD, dtype=torch.int32, device=self.device)
        tge_dev[:, :target_l] = targets_eval
        edist, tlen = edit_distance_device(labels, tge_dev, self._stream())
        labels_h = labels.cpu().numpy()
        dist_h, tlen_h = edist.cpu().numpy(), tlen.cpu().numpy()
        word_err = float((dist_h != 0).sum())
        labels_pred, labels_gold = [], []
        if self.visualize:
            _, labels_pred, labels_gold = eval_word_err_rate(labels_h, tge_dev.cpu().numpy(), True)
        accuracy = B - word_err
        self._dec_out = SimpleNamespace(labels=labels_h, scores=scores.cpu().numpy(), gold_scores=gold.cpu().numpy(),
                                        edit_distance=dist_h, target_len=tlen_h)
        if self.visualize and self.visualize_file is not None:
            img_paths = batch[4]
            for i in range(len(img_paths)):                                         # model.lua:628-633
                self.visualize_file.write("%s\t%s\t%s\t%f\t%f\n" % (img_paths[i], labels_gold[i], labels_pred[i],
                                                                    self._dec_out.scores[i], self._dec_out.gold_scores[i]))
            self.visualize_file.flush()
        loss = float(loss_dev.item())
        self.check_health()
        return loss, [num_nonzeros, accuracy]

    def forward_logits(self, batch, training=False):
        """Teacher-forced decoder logits (L,B,V) and NLL sum (parity tap; the pre-LogSoftMax output of output_projector.lua:5)."""
        images, targets, targets_eval = self._upload(batch)
        B, _, _, W = images.shape
        L = targets.shape[1]
        check(lib.aocr_model_set_stream(self._h, self._stream()))
        logits = torch.empty((L, B, self.target_vocab_size), dtype=torch.float32, device=self.device)
        check(lib.aocr_forward_logits(self._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, L, int(training), ptr(logits),
                                      ptr(self._scal[0:1])), "aocr_forward_logits")
        return logits.cpu(), float(self._scal[0].item())

    def train_forward_backward(self, batch, grad_scale=None):
        """feval only (no parameter update): loss sum; gradients stay in self.grad_params."""
        images, targets, targets_eval = self._upload(batch)
        B, _, _, W = images.shape
        L = targets.shape[1]
        check(lib.aocr_model_set_stream(self._h, self._stream()))
        self._arm_dropout()
        check(lib.aocr_train_forward_backward(self._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, L,
                                              (1.0 / B) if grad_scale is None else grad_scale, ptr(self._scal[0:1])),
              "aocr_train_forward_backward")
        return float(self._scal[0].item())

    def cluster_status(self) -> int:
        """0 = healthy; otherwise the code of a whole-sequence kernel that gave up waiting for its group since the last call (read and clear)."""
        code = C.c_int32(0)
        check(lib.aocr_cluster_status(self._h, C.byref(code)), "aocr_cluster_status")
        return int(code.value)

    def check_health(self):
        """Raises if a whole-sequence kernel ever gave up waiting for its group (include/aocr.h: aocr_cluster_status); synchronises."""
        code = C.c_int32(self.cluster_status())
        if code.value != 0:
            raise RuntimeError(f"a cluster kernel timed out waiting for its group (code {code.value}): the results of that step are invalid")

    def _arm_dropout(self):
        """nn.Dropout(p) of LSTM.lua:68-69,116-118 for the coming training step (rank-dependent seed under data parallelism)."""
        if self.dropout > 0.0 or getattr(self, "_dropout_armed", False):
            # mask counter = global_step (train.lua:115 advances it in the host loop) + 2^32 x (training steps already taken at this
            # global_step): a caller that never advances global_step still draws a fresh mask every step, and the first step at a
            # given global_step uses exactly (seed, global_step) -- what the oracle replays (oracle_torch.dropout_state)
            gs = int(self.global_step)
            rep = self._drop_repeat + 1 if gs == self._drop_last_gs else 0
            self._drop_last_gs, self._drop_repeat = gs, rep
            check(lib.aocr_set_dropout(self._h, self.dropout, self.dropout_seed + 7919 * dist.rank(), gs + (rep << 32)), "aocr_set_dropout")
            self._dropout_armed = self.dropout > 0.0

    def sgd_step(self, lr=None, clip=5.0):
        norms = self._scal[2:12]
        check(lib.aocr_sgd_step(self._h, float(self.optim_state["learningRate"] if lr is None else lr), clip, ptr(norms)))
        return norms.cpu().numpy().reshape(5, 2)

    def format_norms(self):
        """the lines optim_sgd.lua:49 prints every step, from the norms the last step() left on the device; reading them is the only
        host sync, so the reference's per-step print becomes an on-demand report."""
        if self.last_norms is None:
            return []
        n = self.last_norms.cpu().numpy().reshape(5, 2)
        return ["i: %d, param norm: %f, grad norm: %f" % (i + 1, n[i, 0], n[i, 1]) for i in range(5)]

    def adadelta_step(self, rho=0.9, eps=1e-6, weight_decay=0.0):
        """optim.adadelta_list (src/optim/optim_adadelta.lua:19-62) on the gradients of the last train_forward_backward; the state
        {paramVariance | accDelta} lives in `self.adadelta_state` (2 x num_params floats on the device)."""
        if getattr(self, "adadelta_state", None) is None:
            self.adadelta_state = torch.zeros(2 * self.num_params, dtype=torch.float32, device=self.device)
        check(lib.aocr_adadelta_step(self._h, rho, eps, weight_decay, ptr(self.adadelta_state)), "aocr_adadelta_step")

    # ------------------------------------------------------------------ misc API of the reference class
    def vis(self, output_dir):
        """model:vis, model.lua:708-718."""
        self.visualize = True
        self.visualize_path = os.path.join(output_dir, "results.txt")
        try:
            self.visualize_file = open(self.visualize_path, "w")
        except OSError:
            print(f"Error: visualize file {self.visualize_path} cannot be created")
            self.visualize = False
            self.visualize_file = None

    def save(self, model_path, layout=None):
        """model:save, model.lua:720-725 ({nets, config, global_step, optim_state}).
        layout None: a torch.save file with the flat parameter vector (fast path for this package);  a path ending in .t7 is written
        in Torch7 serialization: layout "flat" (the default for .t7) = one plain Lua table of named FloatTensors (write_flat_checkpoint:
        depends on nothing but torch.load of a table; INTEGRATION.md shows the Lua that pours it into a model the reference has just
        created), layout "reference" (opt-in) = the reference's own table with the five nets as nn / nngraph object trees
        (aocr.checkpoint.write_reference_checkpoint).  The object trees are UNVERIFIED against Torch7 -- nngraph's gModule is an
        nn.Container whose parameters() / type() walk self.modules and whose forward / backward use fg, bg, innode, outnode and
        backwardnodes, none of which can be reproduced faithfully without the un-vendored packages -- so that layout is never what a
        caller gets without asking for it.
        No collective happens here (train.lua saves from one process): under data parallelism WITHOUT synchronised BatchNorm the
        running statistics are rank-local -- every rank calls `sync_bn_state()` before rank 0 saves."""
        if str(model_path).endswith(".t7") or layout in ("reference", "flat"):
            from .checkpoint import write_flat_checkpoint, write_reference_checkpoint
            if layout is None:                                    # the .t7 default is NOT what upstream model:load reads -- say so once per call
                import warnings
                warnings.warn(f"Model.save({str(model_path)!r}): writing the flat 'aocr-flat-1' table; the reference's model:load "
                              "(src/model/model.lua:52-60) reads layout='reference' (or the Lua glue of INTEGRATION.md)", stacklevel=2)
            writer = write_reference_checkpoint if layout == "reference" else write_flat_checkpoint
            writer(model_path, {k: v.numpy() for k, v in self.get_parameters().items()},
                   {k: v.numpy() for k, v in self.get_bn_state().items()}, self.config, self.global_step, self.optim_state)
            return
        torch.save({"params": self.params.detach().cpu(), "bn_state": self.bn_state.detach().cpu(), "config": self.config,
                    "global_step": self.global_step, "optim_state": dict(self.optim_state)}, model_path)

    def sync_bn_state(self):
        """COLLECTIVE (every rank calls it): averages the BatchNorm running statistics over the ranks.  Needed only when the step ran
        with rank-local batch statistics; with synchronised BatchNorm (the default of aocr.dist.attach / attach_rccl) the running
        statistics are already identical on all ranks and this returns without communicating."""
        if dist.world_size() <= 1 or self.sync_bn_active():
            return
        torch.distributed.all_reduce(self.bn_state); self.bn_state.div_(dist.world_size())

    def comm_exposed_ms(self) -> float:
        """ms the model's stream waited for the gradient exchange at the end of the last step (include/aocr.h: aocr_comm_exposed_ms)."""
        ms = C.c_float(0.0)
        check(lib.aocr_comm_exposed_ms(self._h, C.byref(ms)), "aocr_comm_exposed_ms")
        return float(ms.value)

    def sync_bn_active(self) -> bool:
        """True when the library normalises with the statistics of the global batch (asked of the library, include/aocr.h)."""
        n, sb, prov = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.aocr_comm_info(self._h, C.byref(n), C.byref(sb), C.byref(prov)), "aocr_comm_info")
        return bool(sb.value) and prov.value != 0

    def shutdown(self):
        if self.visualize_file:
            self.visualize_file.close()
            self.visualize_file = None
        if self._h is not None:
            lib.aocr_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            if self._h is not None:
                lib.aocr_model_destroy(self._h)
                self._h = None
        except Exception:
            pass
