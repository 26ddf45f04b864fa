"""Data parallelism inside the library (include/aocr.h: aocr_comm_*, aocr_allreduce_grads, synchronised BatchNorm).

* DP-2 == 1 GPU in TRAINING mode: two processes (gloo, both on the one GPU of the box -- RCCL refuses two ranks on one device), each
  with half of a batch, exchange through the library's callback provider with synchronised BatchNorm; loss, every gradient, the
  updated parameters and the running statistics must equal the single-process step on the whole batch.
* The RCCL provider itself (librccl bound by the library with dlopen): a 1-rank communicator on the one GPU -- ncclCommInitRank,
  ncclAllReduce on the library's second stream behind the gradient-ready events -- must leave the step unchanged."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(enc_hidden=32, enc_layers=1, dec_layers=2, input_feed=True)


def _build(B, W, compute="f32"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_step_gpu import make
    return make(CFG, B=B, W=W, maxlen=6, compute=compute, max_decoder_l=8, max_beam=1)


def _worker(rank, world, port, q, sync_bn):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for p in (os.path.join(ROOT, "torch-attention-ocr_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if not sync_bn:
        os.environ["AOCR_NO_SYNC_BN"] = "1"
    os.environ["AOCR_COMM_LOG"] = "1"
    m, O, ocfg, P, st, batch = _build(8, 40)                      # the GLOBAL batch: every rank generates it, then keeps its slice
    half = 8 // world
    sl = slice(rank * half, (rank + 1) * half)
    local = [np.asarray(batch[0])[sl], np.asarray(batch[1])[sl], np.asarray(batch[2])[sl], batch[3], None]
    loss, _ = m.step(local, False)                                  # feval + exchange + clip + update, through Model.step
    grads = {k: v.numpy() for k, v in m.get_gradients().items()}
    params = {k: v.numpy() for k, v in m.get_parameters().items()}
    bn = {k: v.numpy() for k, v in m.get_bn_state().items()}
    q.put((rank, loss, grads, params, bn, list(m._comm_log or [])))
    dist.barrier()
    m.shutdown()
    dist.destroy_process_group()


@pytest.mark.parametrize("sync_bn", [True, False])
def test_dp2_equals_single_gpu_in_training_mode(cuda, sync_bn):
    m, O, ocfg, P, st, batch = _build(8, 40)
    loss1, _ = m.step(batch, False)
    g1 = {k: v.numpy() for k, v in m.get_gradients().items()}
    p1 = {k: v.numpy() for k, v in m.get_parameters().items()}
    b1 = {k: v.numpy() for k, v in m.get_bn_state().items()}
    m.shutdown()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300 + (7 if sync_bn else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sync_bn)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, g0, p0, bn0, log0), (_, l1, gr1, pr1, bnr1, log1) = res
    # the exchange as the library issued it: identical on both ranks (a mismatch would deadlock RCCL), the BatchNorm sums on their own
    # channel (fp64, AOCR_COMM_CHANNEL_BN) -- 3 layers x (forward + backward) -- the four gradient buckets + the loss on channel 0 (fp32)
    assert log0 == log1 and len(log0) > 0
    grads_ch = [e for e in log0 if e[0] == 0]; bn_ch = [e for e in log0 if e[0] == 1]
    # (a model that carries whole-sequence kernels -- bf16 mode -- also sums its time-out flag: one more 1-element entry)
    assert len(grads_ch) in (5, 6) and all(e[1] == 0 for e in grads_ch) and sum(1 for e in grads_ch if e[2] == 1) == len(grads_ch) - 4
    assert (len(bn_ch) == 6 and all(e[1] == 1 for e in bn_ch)) if sync_bn else not bn_ch
    for k in g0:                                   # both ranks hold the same summed gradients and the same updated parameters
        assert np.array_equal(g0[k], gr1[k]) and np.array_equal(p0[k], pr1[k]), k
    assert l0 == pytest.approx(l1)
    noisy = ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b")

    def rel(a, b):
        return float(np.abs(a.astype(np.float64) - b).max() / (np.abs(b).max() + 1e-12))
    worst = max(rel(g0[k], g1[k]) for k in g1 if k not in noisy)
    worst_p = max(float(np.abs(p0[k] - p1[k]).max()) for k in p1)
    worst_bn = max(float(np.abs(bn0[k] - b1[k]).max()) for k in b1)
    print(f"[dp] sync_bn={sync_bn}: loss {l0:.5f} vs single {loss1:.5f}; worst gradient rel {worst:.2e}, parameter max-abs {worst_p:.2e}, running stats {worst_bn:.2e}")
    if sync_bn:                                    # DP-2 == 1 GPU on the concatenated batch, training-mode BatchNorm included
        assert l0 == pytest.approx(loss1, rel=1e-5)
        assert worst < 2e-4 and worst_p < 1e-5 and worst_bn < 1e-5
    else:                                          # per-rank statistics: a different (documented) computation -- must NOT coincide
        assert worst > 1e-3


CFG_BF16 = dict(enc_hidden=256, enc_layers=2, dec_layers=2, input_feed=True)
ENV_BF16 = dict(AOCR_FORCE_DMA="1", AOCR_LAYER_PIPE_CHUNKS="3")          # small batch: still the 256 x 256 conv kernels (staged tiles, BatchNorm sums in the epilogue) and the layer wavefront


def _build_bf16(B, W):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_step_gpu import make
    return make(CFG_BF16, B=B, W=W, maxlen=6, compute="bf16", max_decoder_l=8, max_beam=1)


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", **ENV_BF16)
    for p in (os.path.join(ROOT, "torch-attention-ocr_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, O, ocfg, P, st, batch = _build_bf16(8, 128)
    half = 8 // world
    sl = slice(rank * half, (rank + 1) * half)
    local = [np.asarray(batch[0])[sl], np.asarray(batch[1])[sl], np.asarray(batch[2])[sl], batch[3], None]
    losses = [m.step(local, False)[0] for _ in range(3)]            # three steps: events / epochs / communicator channels are reused
    grads = {k: v.numpy() for k, v in m.get_gradients().items()}
    params = {k: v.numpy() for k, v in m.get_parameters().items()}
    status = int(m.get_tensor("cl_err").view(torch.int32)[0])
    q.put((rank, losses, grads, params, status))
    dist.barrier()
    m.shutdown()
    dist.destroy_process_group()


def test_dp2_bf16_production_dispatch(cuda, monkeypatch):
    """Two ranks (callback provider, synchronised BatchNorm) in bf16 mode with the kernels the benchmark runs: cluster encoder / decoder
    kernels, a stacked encoder as a layer wavefront on its own streams, the decoder's weight gradients on the side stream (the gradient
    bucket events are recorded there), the 4-wave conv kernels with staged tiles and the BatchNorm partial sums from their epilogue
    feeding the synchronised statistics.  Both ranks must end with identical parameters; against the single-process step on the whole
    batch only the summation order differs (bf16 tolerances)."""
    for k, v in ENV_BF16.items():
        monkeypatch.setenv(k, v)
    m, O, ocfg, P, st, batch = _build_bf16(8, 128)
    l_single = [m.step(batch, False)[0] for _ in range(3)]
    g1 = {k: v.numpy() for k, v in m.get_gradients().items()}
    p1 = {k: v.numpy() for k, v in m.get_parameters().items()}
    m.shutdown()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_worker_bf16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, g0, p0, s0), (_, l1, gr1, pr1, s1) = res
    assert s0 == 0 and s1 == 0, "a cluster kernel timed out"
    for k in p0:
        assert np.array_equal(p0[k], pr1[k]) and np.array_equal(g0[k], gr1[k]), k
    assert l0 == pytest.approx(l1)

    def cos(a, b):
        a = a.astype(np.float64).ravel(); b = b.astype(np.float64).ravel()
        return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
    allc = sorted((cos(g0[k], g1[k]), k) for k in g1 if k not in ("cnn.conv3.b", "cnn.conv5.b", "cnn.conv7.b"))
    print("[dp] lowest gradient cosines (after the third step: the parameters of the two runs have drifted apart by then):", [(round(c, 5), k) for c, k in allc[:6]])
    worst = allc[0]
    dp = max(float(np.abs(p0[k] - p1[k]).max()) for k in p1)
    print(f"[dp] bf16, production dispatch: losses {[round(x, 4) for x in l0]} vs single {[round(x, 4) for x in l_single]}; worst gradient cosine {worst[0]:.6f} ({worst[1]}), parameter max-abs {dp:.2e}")
    for a, b in zip(l0, l_single):
        assert a == pytest.approx(b, rel=5e-3)
    assert worst[0] > 0.9 and allc[len(allc) // 2][0] > 0.995, allc[:4]      # 8 images in bf16: the small bias vectors are noisy (the fp64 oracle sees them at 0.98 too, test_halo_four_wave_...)


def _worker_timeout(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0", **ENV_BF16)
    for p in (os.path.join(ROOT, "torch-attention-ocr_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, O, ocfg, P, st, batch = _build_bf16(8, 128)
    half = 8 // world
    sl = slice(rank * half, (rank + 1) * half)
    local = [np.asarray(batch[0])[sl], np.asarray(batch[1])[sl], np.asarray(batch[2])[sl], batch[3], None]
    m.set_parameters(P, st)
    m.forward_logits(local, training=False)                        # (the parity taps -- the status word among them -- exist once a step has run; evaluation mode: nothing moves)
    p0 = m.params.clone(); bn0 = {k: v.clone() for k, v in m.get_bn_state().items()}
    flag = m.get_tensor_view("cl_err")[:1]
    images, targets, targets_eval = m._upload(local)
    if rank == 1:
        # THIS rank's encoder / decoder kernel "times out" DURING the step: the code is injected right behind feval, in stream order, in front of the exchange (as in
        # test_update_is_skipped_...).  (Round 6: a code that is already there when a training step STARTS is a stale one -- a decode call's -- and is moved aside
        # by the step's prologue, include/aocr.h: aocr_cluster_status; injected before the call it would no longer model a time-out of this step.)
        import aocr.model as AM
        orig = AM.lib.aocr_train_forward_backward
        armed = [True]
        def feval_then_time_out(*a):
            r = orig(*a)
            if armed[0]:
                armed[0] = False; flag.fill_(23)
            return r
        AM.lib.aocr_train_forward_backward = feval_then_time_out
    m.train_step_device(images, targets, targets_eval, 8)           # feval + exchange (the flag travels with it) + clip + update
    torch.cuda.synchronize()
    skipped = bool(torch.equal(m.params, p0))
    code = m.cluster_status()                                       # read and clear
    bn_mid = {k: v.clone() for k, v in m.get_bn_state().items()}
    m.train_step_device(images, targets, targets_eval, 8)           # the repeat, on every rank together
    torch.cuda.synchronize()
    code2 = m.cluster_status()
    params = {k: v.numpy() for k, v in m.get_parameters().items()}
    bn = {k: v.numpy() for k, v in m.get_bn_state().items()}
    same_bn = all(torch.equal(bn_mid[k], bn0[k]) for k in bn0)       # the skipped step's move was taken back by the optimizer call that skipped (round 5)
    moved_bn = any(not torch.equal(bn_mid[k], torch.from_numpy(bn[k])) for k in bn)      # ... and the repeat moved them (once: compared with the clean step below)
    q.put((rank, skipped, code, code2, params, bn, same_bn, moved_bn))
    dist.barrier()
    m.shutdown()
    dist.destroy_process_group()


def test_dp2_cluster_timeout_is_a_global_decision(cuda, monkeypatch):
    """a whole-sequence kernel that timed out on ONE rank must make EVERY rank skip the update (and repeat the step
    together): the time-out flag is summed with the exchange (comm.hip), so the peers' optimizer predicate sees it too.  Rank 1 injects a
    code; both ranks must keep their parameters, both must report a non-zero status (rank 0: the peer code 0x7e), the repeated step must
    leave both with the parameters of ONE clean step, and the BatchNorm running statistics must have moved exactly once."""
    for k, v in ENV_BF16.items():
        monkeypatch.setenv(k, v)
    m, O, ocfg, P, st, batch = _build_bf16(8, 128)
    m.step(batch, False)                                            # the clean single-process step on the whole batch
    p1 = {k: v.numpy() for k, v in m.get_parameters().items()}
    b1 = {k: v.numpy() for k, v in m.get_bn_state().items()}
    m.shutdown()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + os.getpid() % 40
    procs = [ctx.Process(target=_worker_timeout, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, sk0, c0, c0b, pa0, bna0, same0, moved0), (_, sk1, c1, c1b, pa1, bna1, same1, moved1) = res
    print(f"[dp] injected time-out on rank 1: update skipped on rank 0 / 1: {sk0} / {sk1}; status codes {c0:#x} / {c1:#x}, after the repeat {c0b} / {c1b}")
    assert sk0 and sk1, "a rank applied an update computed from gradients another rank flagged invalid"
    assert c0 == 0x7e and c1 == 23 and c0b == 0 and c1b == 0
    assert same0 and same1 and moved0 and moved1, "the skipped step left its move of the BatchNorm running statistics behind, or the repeat did not move them"
    for k in pa0:
        assert np.array_equal(pa0[k], pa1[k]), k                    # the replicas did not diverge
    dp = max(float(np.abs(pa0[k] - p1[k]).max()) for k in p1)
    db = max(float(np.abs(bna0[k] - b1[k]).max()) for k in b1)
    print(f"[dp] after the repeat: parameter max-abs vs one clean single-process step {dp:.2e}, running statistics {db:.2e}")
    assert dp < 2e-3 and db < 1e-3                                  # bf16 summation order only (DP-2 vs 1 GPU)


def test_rccl_provider_single_rank(cuda):
    """librccl through the library's own binding (ncclGetUniqueId / ncclCommInitRank / ncclAllReduce on the second stream)."""
    import ctypes as C
    import aocr
    from aocr import check, lib, ptr
    from aocr import dist as adist
    m, O, ocfg, P, st, batch = _build(8, 40, compute="bf16")
    loss0 = m.train_forward_backward(batch)
    g0 = m.grad_params.clone()
    adist.attach_rccl(m, sync_bn=True)              # world of 1: every all-reduce is the identity, but goes through RCCL
    images, targets, targets_eval = m._upload(batch)
    B, _, _, W = images.shape
    loss = torch.zeros(1, device=cuda)
    for _ in range(3):
        check(lib.aocr_train_forward_backward(m._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, targets.shape[1], 1.0 / B, ptr(loss)))
        check(lib.aocr_allreduce_grads(m._h, ptr(loss)), "aocr_allreduce_grads")
        g = m.grad_params.clone()                   # on the model's stream, which has joined the exchange stream
        torch.cuda.synchronize()
        assert (g - g0).abs().max().item() <= 1e-4 * g0.abs().max().item()
        assert loss.item() == pytest.approx(loss0, rel=1e-5)
    check(lib.aocr_comm_destroy(m._h))
    m.shutdown()


def test_attach_is_a_no_op_after_attach_rccl(cuda, monkeypatch):
    """Model.train_step_device calls dist.attach() on every step of a multi-rank run; on a model that already carries
    the library's own RCCL provider (attach_rccl) that must return early instead of tripping aocr_comm_set_callback's
    'already attached' error, and a failed attach must not leave `_comm_cb` set."""
    from aocr import dist as adist
    m, O, ocfg, P, st, batch = _build(4, 40, compute="f32")
    adist.attach_rccl(m, sync_bn=True)                                    # 1-rank communicator
    monkeypatch.setattr(adist, "world_size", lambda: 2)                   # as seen from inside a 2-rank job
    adist.attach(m)                                                       # no error, nothing attached on top
    assert getattr(m, "_comm_cb", None) is None and m._comm_rccl
    assert m.sync_bn_active()
    loss = m.train_forward_backward(batch)                                # the 1-rank communicator still works (sums are identities)
    assert np.isfinite(loss)
    m.shutdown()
