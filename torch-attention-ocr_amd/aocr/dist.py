"""Data parallelism over the image batch: the ONE exchange step of the hot path.

The reference is single-GPU (src/train.lua:246).  Every sample is independent except the BatchNorm batch
statistics and the scalar loss, so each rank runs the full step on its slice with d(loss) scaled by
1/global_batch (the reference divides by the step's batch size, model.lua:645-647) and the ranks sum their flat
gradient vectors with one all-reduce placed between feval and the per-group clip (optim_sgd.lua:38 -> :40), so
every rank clips and updates identically.  `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm; the
same code runs on gloo for the CPU tests.  BatchNorm statistics stay per-rank (documented deviation).

The flat gradient vector completes back to front during the backward pass (decoder + projector, encoders, CNN from conv7
down); `exchange_overlapped` sums it in the four buckets the library marks with events (`aocr_grad_buckets`,
`aocr_stream_wait_grads`) on a second stream, so that on xGMI -- point-to-point links, a ring all-reduce of 48.7 MB costs
about a millisecond at 8 GPUs -- only the last 3.8 MB bucket (conv1..conv4) is exposed after the backward pass.
"""
from __future__ import annotations

import ctypes as C
import os

import torch


def world_size() -> int:
    d = torch.distributed
    return d.get_world_size() if (d.is_available() and d.is_initialized()) else 1


def rank() -> int:
    d = torch.distributed
    return d.get_rank() if (d.is_available() and d.is_initialized()) else 0


def grad_scale(local_batch: int) -> float:
    """d(loss)/d(pred) scale so that the SUM over ranks equals the reference's 1/batch_size of the global batch."""
    return 1.0 / (local_batch * world_size())


def exchange(flat_grads: torch.Tensor, loss: torch.Tensor) -> None:
    """In-place sum over ranks of the flat gradient vector (one bucket: 12.18 M fp32 = 48.7 MB at He=256) and the loss."""
    if world_size() > 1:
        torch.distributed.all_reduce(flat_grads)
        torch.distributed.all_reduce(loss)


def bucket_ranges(cfg):
    """[(begin, end)] float ranges of the gradient buckets in completion order (`aocr_grad_buckets`)."""
    import ctypes as C
    from ._lib import check, lib
    b, e = (C.c_int64 * 4)(), (C.c_int64 * 4)()
    check(lib.aocr_grad_buckets(C.byref(cfg), b, e), "aocr_grad_buckets")
    return [(int(b[i]), int(e[i])) for i in range(4)]


def exchange_overlapped(flat_grads: torch.Tensor, loss: torch.Tensor, ranges, wait_bucket, comm_stream=None) -> None:
    """Bucketed form of `exchange`: for each bucket k (in completion order) `wait_bucket(k, stream)` makes the communication
    stream wait for the bucket's gradient-ready event, then the bucket is summed over ranks; the caller's stream joins the
    communication stream at the end.  On CPU tensors (gloo tests) `comm_stream` is None and the buckets are summed in order."""
    if world_size() <= 1:
        return
    d = torch.distributed
    if comm_stream is None:
        for k, (b, e) in enumerate(ranges):
            wait_bucket(k, None)
            d.all_reduce(flat_grads[b:e])
        d.all_reduce(loss)
        return
    main = torch.cuda.current_stream(flat_grads.device)
    with torch.cuda.stream(comm_stream):
        for k, (b, e) in enumerate(ranges):
            wait_bucket(k, comm_stream)
            d.all_reduce(flat_grads[b:e])            # enqueued behind the event; returns without a host sync
            if k == 0:
                d.all_reduce(loss)                   # the loss is final before the backward pass starts
    main.wait_stream(comm_stream)                    # the clip + update need every bucket


# ------------------------------------------------------------------------------------------------ the exchange inside the library
def attach(model, sync_bn: bool = True) -> None:
    """Hands torch.distributed to the library as its all-reduce provider (`aocr_comm_set_callback`, include/aocr.h): from then on
    `aocr_allreduce_grads` sums the gradient buckets on the library's own second stream beside the backward pass, and -- with
    sync_bn -- the three BatchNorm layers normalise with the statistics of the GLOBAL batch, so that N ranks on slices of a batch
    compute what one GPU computes on the whole batch.  A Lua / C host uses `aocr_comm_init_rank` (RCCL bound by the library) instead."""
    from ._lib import ALLREDUCE_FN, check, lib
    if world_size() <= 1 or getattr(model, "_comm_cb", None) is not None or getattr(model, "_comm_rccl", False):
        return                                       # already attached (callback or the library's own RCCL provider)
    d = torch.distributed
    # the BatchNorm sums travel on a process group of their own (include/aocr.h, AOCR_COMM_CHANNEL_BN): a group runs its collectives in
    # issue order, and the gradient buckets -- issued after the whole backward pass has been enqueued -- must not queue behind the last
    # BatchNorm-backward sum.  new_group is collective: every rank attaches at the same point (the first training step).
    # new_group is collective, so it must not hang on a per-rank difference: every rank creates the group, and the two switches that decide
    # whether it is USED (sync_bn, AOCR_ONE_COMM) are first agreed on -- a rank whose switches differ from its peers' raises instead of
    # summing its BatchNorm statistics on a different group than they do (mismatched collectives: a hang at the first sum).
    mine = torch.tensor([float(bool(sync_bn)), float(bool(os.environ.get("AOCR_ONE_COMM")))], dtype=torch.float32,
                        device=model.device if d.get_backend() == "nccl" else "cpu")       # nccl (= RCCL) sums device tensors only
    lo, hi = mine.clone(), mine.clone()
    d.all_reduce(lo, op=d.ReduceOp.MIN); d.all_reduce(hi, op=d.ReduceOp.MAX)
    if not (torch.equal(lo, mine) and torch.equal(hi, mine)):
        raise RuntimeError(f"aocr.dist.attach: sync_bn / AOCR_ONE_COMM differ between the ranks (this rank {mine.tolist()}, min {lo.tolist()}, max {hi.tolist()})")
    group = d.new_group()
    model._bn_group = group if (sync_bn and not os.environ.get("AOCR_ONE_COMM")) else None
    model._comm_log = [] if os.environ.get("AOCR_COMM_LOG") else None

    def view(ptr, count, dtype):
        nbytes = count * (8 if dtype else 4)
        for t in (model.grad_params, model.workspace, model._scal):
            base = t.data_ptr()
            if base <= ptr and ptr + nbytes <= base + t.numel() * t.element_size():
                off = ptr - base
                flat = t.view(torch.uint8).reshape(-1)[off:off + nbytes]
                return flat.view(torch.float64 if dtype else torch.float32)
        raise RuntimeError("all-reduce of a buffer outside the model's tensors")

    def cb(user, buf, count, dtype, stream):
        try:
            chan, dtype = dtype >> 8, dtype & 0xFF
            v = view(buf, count, dtype)
            if model._comm_log is not None:
                model._comm_log.append((chan, dtype, int(count)))
            ext = torch.cuda.ExternalStream(stream, device=model.device) if stream else torch.cuda.default_stream(model.device)
            with torch.cuda.stream(ext):
                d.all_reduce(v, group=model._bn_group if chan else None)
            return 0
        except Exception as e:                       # never let an exception cross the C boundary
            print(f"[aocr.dist] all-reduce callback failed: {e!r}", flush=True)
            return 1
    tramp = ALLREDUCE_FN(cb)
    check(lib.aocr_comm_set_callback(model._h, C.cast(tramp, C.c_void_p), None, world_size(), int(sync_bn)), "aocr_comm_set_callback")
    model._comm_cb = tramp                           # only once the library took it; keeps the trampoline alive as long as the model


def attach_rccl(model, sync_bn: bool = True) -> None:
    """The library's own RCCL provider (what a non-Python host uses): rank 0's unique id travels through torch.distributed's store."""
    from ._lib import check, lib
    d = torch.distributed
    n, r = (d.get_world_size(), d.get_rank()) if (d.is_available() and d.is_initialized()) else (1, 0)
    buf = C.create_string_buffer(128)
    if r == 0:
        check(lib.aocr_comm_unique_id(buf), "aocr_comm_unique_id")
    if n > 1:
        ids = [bytes(buf.raw)]
        d.broadcast_object_list(ids, 0)
        buf = C.create_string_buffer(ids[0], 128)
    check(lib.aocr_comm_init_rank(model._h, buf, n, r, int(sync_bn)), "aocr_comm_init_rank")
    model._comm_rccl = True
