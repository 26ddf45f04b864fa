"""The exchange and the whole-sequence ("cluster") kernels must not collide.

A collective's kernel stays resident until every peer has joined it; a cluster kernel needs all members of a group resident at once and
bounds its spins (aocr_cluster_status).  Nothing in this pool has more than one GPU, so the collective is played by a CO-TENANT kernel
with its footprint -- N workgroups of 512 threads that stream a buffer and stay resident for a fixed time (tests/cotenant.hip) -- which
the all-reduce CALLBACK launches on the library's exchange stream whenever a gradient bucket is handed to it: exactly where and when an
RCCL all-reduce of that bucket would start.

* default policy: bucket 0 is held behind the encoder BPTT -> no co-tenant is ever beside a cluster kernel;
* AOCR_COMM_EARLY_BUCKET0=1: bucket 0 is released as soon as the decoder's gradients are complete -> the co-tenant runs ACROSS
  enc_cl_bwd_kernel, which then leaves AOCR_COMM_RESERVE_CUS (default 32) compute units free;
* the same with AOCR_COMM_RESERVE_CUS=0 at He = 512, batch 256 (the encoder kernel wants all 256 compute units): the members placed first
  spin until the co-tenant leaves -- the hazard the reserve removes; still no time-out, because the co-tenant leaves after a millisecond.

Every variant must end with aocr_cluster_status == 0 and the parameters of the undisturbed run; step times are reported.
Also here: the library's own RCCL provider driven through Model.step (1-rank communicator: ncclCommInitRank, ncclCommSplit, ncclAllReduce)."""
import ctypes as C
import os
import time

import numpy as np
import pytest
import torch

from test_step_gpu import make

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _cotenant():
    p = os.path.join(HERE, "libcotenant.so")
    if not os.path.exists(p):
        pytest.fail("tests/libcotenant.so is missing: run __graft_entry__.build() (make -C tests -f Makefile.harness all)")
    lib = C.CDLL(p)
    lib.cotenant_launch.restype = C.c_int
    lib.cotenant_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_double, C.c_void_p]
    return lib


def _attach_cotenant(m, co, wgs, usec, log):
    """Callback provider of a 2-"rank" job whose peer contributes zeros: the sum is the identity, and every gradient bucket (channel 0,
    more than 1000 elements) starts a co-tenant on the exchange stream."""
    from aocr import check, lib
    from aocr._lib import ALLREDUCE_FN
    scratch = torch.zeros(64 << 20, dtype=torch.uint8, device=m.device)
    stamps = torch.zeros(2 * wgs, dtype=torch.int64, device=m.device)

    def cb(user, buf, count, dtype, stream):
        chan = dtype >> 8
        if chan == 0 and count > 1000:
            log.append(int(count))
            rc = co.cotenant_launch(stream, scratch.data_ptr(), scratch.numel(), wgs, float(usec), stamps.data_ptr())
            return 0 if rc == 0 else 1
        return 0
    tramp = ALLREDUCE_FN(cb)
    check(lib.aocr_comm_set_callback(m._h, C.cast(tramp, C.c_void_p), None, 2, 1), "aocr_comm_set_callback")
    m._comm_cb = tramp
    m._keep = (scratch, stamps)


def _steps(m, batch, n=4):
    images, targets, targets_eval = m._upload(batch)
    B = images.shape[0]
    losses, times = [], []
    for i in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ld = m.train_step_device(images, targets, targets_eval, B)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        losses.append(float(ld.item()))
    return losses, min(times[1:])


@pytest.mark.parametrize("He,B,W", [(256, 256, 100), (512, 256, 100)])
def test_cotenant_across_encoder_bptt(cuda, monkeypatch, He, B, W):
    co = _cotenant()
    cfg = dict(enc_hidden=He, enc_layers=1, dec_layers=2, input_feed=True)
    WGS, USEC = 32, 1000.0
    results = {}
    for tag, env in (("plain", None), ("hold", {}), ("early+reserve", {"AOCR_COMM_EARLY_BUCKET0": "1"}),
                     ("early, no reserve", {"AOCR_COMM_EARLY_BUCKET0": "1", "AOCR_COMM_RESERVE_CUS": "0"})):
        for k in ("AOCR_COMM_EARLY_BUCKET0", "AOCR_COMM_RESERVE_CUS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=11, compute="bf16", max_decoder_l=12, max_beam=1)
        log = []
        if env is not None:
            _attach_cotenant(m, co, WGS, USEC, log)
        losses, ms = _steps(m, batch)
        status = m.cluster_status()
        params = m.params.clone()
        results[tag] = (losses, ms, status, params, len(log))
        print(f"[cotenant] He={He} B={B} {tag:18s}: {ms:7.3f} ms/step, cluster status {status}, losses {[round(x, 3) for x in losses]}, co-tenant launches {len(log)}")
        assert status == 0, (tag, status)
        m.shutdown()
    ref_losses, ref_ms, _, ref_params, _ = results["plain"]
    for tag, (losses, ms, status, params, nlog) in results.items():
        if tag == "plain":
            continue
        assert nlog == 4 * 4, (tag, nlog)                        # four buckets per step, four steps
        for a, b in zip(losses, ref_losses):
            assert a == pytest.approx(b, rel=2e-3), (tag, losses, ref_losses)
        d = (params - ref_params).abs().max().item()
        assert d < 5e-3, (tag, d)                                # the same training run (bf16 summation order of the split-K atomics only)
    # the co-tenants of one step hold 32 compute units for 4 x 1 ms on the exchange stream; the backward pass they overlap is ~2-4 ms,
    # so a step may become longer -- but never by more than the co-tenants' own time
    for tag in ("hold", "early+reserve", "early, no reserve"):
        assert results[tag][1] < ref_ms + 4 * USEC * 1e-3 + 1.0, (tag, results[tag][1], ref_ms)


@pytest.mark.parametrize("WGS", [32, 64])
def test_cotenant_across_cnn_backward(cuda, WGS):
    """with bucket 0 held behind the encoder BPTT (the default), the exchange lives in the CNN backward window,
    where the data gradients and the filter gradients already co-run on two streams -- the collective is the THIRD tenant there.  A co-tenant
    with RCCL's footprint (32 or 64 workgroups x 512 threads, copy loop) is started by the bucket-0 all-reduce and stays for 2.5 ms = the whole
    CNN backward at the C3 shape.  The gradients must be those of the undisturbed step (the kernels' grids are sized by tiles, not by
    free compute units: a tenant can only delay them), every bucket must still be handed over, and the step-time delta is reported."""
    from aocr import check, lib, ptr
    co = _cotenant()
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    USEC = 2500.0
    out = {}
    for tag in ("plain", "cotenant"):
        m, O, ocfg, P, st, batch = make(cfg, B=256, W=256, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=1)
        log = []
        if tag == "cotenant":
            from aocr._lib import ALLREDUCE_FN
            scratch = torch.zeros(64 << 20, dtype=torch.uint8, device=m.device)
            stamps = torch.zeros(2 * WGS, dtype=torch.int64, device=m.device)

            def cb(user, buf, count, dtype, stream):             # ONE long co-tenant per step, started where bucket 0's all-reduce starts (peer contributes zeros: the sum is the identity)
                if (dtype >> 8) == 0 and count > 1000:
                    log.append(int(count))
                    if len(log) % 4 == 1:
                        return 0 if co.cotenant_launch(stream, scratch.data_ptr(), scratch.numel(), WGS, float(USEC), stamps.data_ptr()) == 0 else 1
                return 0
            tramp = ALLREDUCE_FN(cb)
            check(lib.aocr_comm_set_callback(m._h, C.cast(tramp, C.c_void_p), None, 2, 1), "aocr_comm_set_callback")
            m._comm_cb = tramp; m._keep = (scratch, stamps)
        images, targets, targets_eval = m._upload(batch)
        loss = torch.zeros(1, device=cuda)
        times = []
        for i in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            check(lib.aocr_train_forward_backward(m._h, ptr(images), ptr(targets), ptr(targets_eval), 256, 256, targets.shape[1], 1.0 / 256, ptr(loss)))
            if tag == "cotenant":
                check(lib.aocr_allreduce_grads(m._h, ptr(loss)), "aocr_allreduce_grads")
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
        out[tag] = (m.grad_params.clone(), float(loss.item()), min(times[1:]), m.cluster_status(), len(log))
        m.shutdown()
    (g0, l0, t0_, s0, _), (g1, l1, t1_, s1, n1) = out["plain"], out["cotenant"]
    print(f"[cotenant] CNN-backward window, {WGS} workgroups x 2.5 ms: feval {t0_:.3f} -> {t1_:.3f} ms (+{t1_ - t0_:.3f}), buckets handed over {n1}")
    assert s0 == 0 and s1 == 0 and n1 == 16
    assert l1 == pytest.approx(l0, rel=1e-5)
    rel = ((g1 - g0).norm() / g0.norm()).item()
    assert rel < 1e-4, rel                                       # split-K partial sums meet in another order under contention: nothing else may change
    assert t1_ < t0_ + USEC * 1e-3 + 1.0                         # never longer than the tenant's own stay


@pytest.mark.parametrize("WGS,USEC", [(32, 4000.0), (96, 1500.0)])
def test_cotenant_across_decoder_chain_kernels(cuda, WGS, USEC):
    """The tag-free exchange of dec_chain.hip under UNEVEN load (MI355X_MICROARCH.md: hand-offs must be tested with part of the chip busy): a co-tenant that
    holds 32 (96) compute units for 4 (1.5) ms is started on another stream right in front of a C3-shape step, so the whole-sequence kernels -- which need every
    compute unit -- run with some members placed late: their peers poll the unwritten pattern for as long as that takes.  The step must end with
    cluster status 0 and with EXACTLY the outputs of the undisturbed step (every exchanged dword either is the pattern or is final: nothing else may be read)."""
    co = _cotenant()
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    m, O, ocfg, P, st, batch = make(cfg, B=256, W=256, maxlen=23, compute="bf16", max_decoder_l=24, max_beam=5)
    taps = ("logits", "outs", "ds_all", "dq_all", "dpre_all")
    m.train_forward_backward(batch)
    ref = {k: m.get_tensor(k).clone() for k in taps}
    scratch = torch.zeros(64 << 20, dtype=torch.uint8, device=m.device)
    stamps = torch.zeros(2 * WGS, dtype=torch.int64, device=m.device)
    side = torch.cuda.Stream(device=m.device)
    for rep in range(3):
        torch.cuda.synchronize()
        assert co.cotenant_launch(side.cuda_stream, scratch.data_ptr(), scratch.numel(), WGS, float(USEC), stamps.data_ptr()) == 0
        m.train_forward_backward(batch)
        torch.cuda.synchronize()
        assert m.cluster_status() == 0
        for k in taps:
            assert torch.equal(m.get_tensor(k), ref[k]), (rep, k)
        lab_ref = m.decode_device(*m._upload(batch), 1)[0].clone()      # (evaluation-mode BatchNorm: the running statistics moved with the training forward above)
        torch.cuda.synchronize()
        assert co.cotenant_launch(side.cuda_stream, scratch.data_ptr(), scratch.numel(), WGS, float(USEC), stamps.data_ptr()) == 0
        lab = m.decode_device(*m._upload(batch), 1)[0]
        torch.cuda.synchronize()
        assert m.cluster_status() == 0 and torch.equal(lab, lab_ref), rep
        # beam search on the chain kernel (six launches of <= 8 groups, token + parent words polled by every member): the same under the co-tenant
        lab5_ref, sc5_ref = (x.clone() for x in m.decode_device(*m._upload(batch), 5)[:2])
        torch.cuda.synchronize()
        assert co.cotenant_launch(side.cuda_stream, scratch.data_ptr(), scratch.numel(), WGS, float(USEC), stamps.data_ptr()) == 0
        lab5, sc5 = m.decode_device(*m._upload(batch), 5)[:2]
        torch.cuda.synchronize()
        assert m.cluster_status() == 0 and torch.equal(lab5, lab5_ref) and torch.equal(sc5, sc5_ref), rep
    m.shutdown()


def test_rccl_provider_through_model_step(cuda):
    """The library's own RCCL binding (ncclCommInitRank + ncclCommSplit for the BatchNorm sums + ncclAllReduce on the exchange stream)
    under Model.step for three optimisation steps on a 1-rank communicator: the sums are identities, so losses and parameters must follow
    the plain run; cluster status 0; the attached provider is reported as RCCL with synchronised BatchNorm."""
    from aocr import check, lib
    from aocr import dist as adist
    cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
    out = {}
    for tag in ("plain", "rccl"):
        m, O, ocfg, P, st, batch = make(cfg, B=32, W=100, maxlen=7, compute="bf16", max_decoder_l=8, max_beam=1)
        if tag == "rccl":
            adist.attach_rccl(m, sync_bn=True)
            n, sb, prov = C.c_int32(), C.c_int32(), C.c_int32()
            check(lib.aocr_comm_info(m._h, C.byref(n), C.byref(sb), C.byref(prov)), "aocr_comm_info")
            assert (n.value, sb.value, prov.value) == (1, 1, 1)
        losses = [m.step(batch, False)[0] for _ in range(3)]
        out[tag] = (losses, m.params.clone(), m.bn_state.clone(), m.cluster_status())
        if tag == "rccl":
            print(f"[rccl] 3 steps through Model.step on a 1-rank communicator: losses {[round(x, 4) for x in losses]}, exposed exchange {m.comm_exposed_ms():.3f} ms")
            check(lib.aocr_comm_destroy(m._h))
        m.shutdown()
    (l0, p0, b0, s0), (l1, p1, b1, s1) = out["plain"], out["rccl"]
    assert s0 == 0 and s1 == 0
    for a, b in zip(l0, l1):
        assert a == pytest.approx(b, rel=2e-3)
    assert (p0 - p1).abs().max().item() < 5e-3 and (b0 - b1).abs().max().item() < 1e-3
