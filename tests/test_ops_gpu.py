"""GPU parity of every module-level C-ABI entry point against plain PyTorch CPU float64
references of the same op (the nn.Module surface of src/model/*.lua).  Tolerances are for
the exact-fp32 MFMA path (v_mfma_f32_32x32x2_f32 = an fp32 fmaf chain)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    import aocr
    return aocr


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


_KEEP = []


def dev(t, dtype=torch.float32):
    """Upload and keep a reference: a.ptr(dev(x)) on a temporary would let the caching allocator recycle the
    block for the next upload before the kernel has run."""
    d = t.to(dtype).cuda().contiguous()
    _KEEP.append(d)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return d


def report(name, got, ref, tol):
    got = got.double().cpu(); ref = ref.double().cpu()
    err = (got - ref).abs().max().item()
    mag = ref.abs().max().item()
    print(f"[parity] {name}: max-abs err {err:.3e} (ref max {mag:.3e}, tol {tol:.1e})")
    assert err <= tol, f"{name}: {err} > {tol}"


@pytest.mark.parametrize("M,N,K,ak,bk", [(96, 80, 64, 1, 1), (200, 39, 512, 1, 1), (130, 512, 39, 1, 0), (39, 300, 777, 0, 0),
                                         (256, 256, 20, 1, 1), (1536, 1024, 512, 1, 1)])
def test_gemm(cuda, M, N, K, ak, bk):
    a = _lib()
    A = rnd(M, K, seed=1); Bm = rnd(N, K, seed=2); bias = rnd(N, seed=3)
    ref = A @ Bm.t() + bias
    Ad = dev(A if ak else A.t()); Bd = dev(Bm if bk else Bm.t())
    Cd = torch.zeros(M, N, device="cuda")
    lda = K if ak else M; ldb = K if bk else N
    a.check(a.lib.aocr_gemm(stream(), 0, a.ptr(Ad), lda, ak, a.ptr(Bd), ldb, bk, a.ptr(Cd), N, M, N, K, a.ptr(dev(bias)), 0))
    report(f"gemm f32 {M}x{N}x{K} ak={ak} bk={bk}", Cd, ref, 2e-5 * max(1, K ** 0.5))
    Cd.zero_()
    a.check(a.lib.aocr_gemm(stream(), 1, a.ptr(Ad), lda, ak, a.ptr(Bd), ldb, bk, a.ptr(Cd), N, M, N, K, a.ptr(dev(bias)), 0))
    report(f"gemm bf16 {M}x{N}x{K} ak={ak} bk={bk}", Cd, ref, 2e-2 * max(1, K ** 0.5))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,pad,relu,pool", [
    (2, 16, 18, 64, 128, 3, 1, 1, 1), (3, 8, 9, 128, 256, 3, 1, 0, 0), (2, 8, 9, 256, 256, 3, 1, 1, 2),
    (2, 2, 9, 512, 512, 2, 0, 0, 0), (1, 4, 25, 512, 512, 3, 1, 1, 2)])
@pytest.mark.parametrize("compute", [0, 1])
def test_conv_forward_backward(cuda, B, H, W, Cin, Cout, ks, pad, relu, pool, compute):
    a = _lib()
    x = rnd(B, Cin, H, W, seed=1).requires_grad_(True)
    w = (rnd(Cout, Cin, ks, ks, seed=2) / (ks * ks * Cin) ** 0.5).requires_grad_(True)
    b = rnd(Cout, seed=3, scale=0.1).requires_grad_(True)
    y0 = F.conv2d(x, w, b, padding=pad)
    y = F.relu(y0) if relu else y0
    if pool == 1:
        y = F.max_pool2d(y, (2, 2), (2, 2))
    elif pool == 2:
        y = F.max_pool2d(y, (2, 1), (2, 1))
    tol = 1e-4 if compute == 0 else 3e-2
    xd = dev(nhwc(x.detach())); wd = dev(w.detach().permute(0, 2, 3, 1)); bd = dev(b.detach())
    yd = torch.zeros(nhwc(y.detach()).shape, device="cuda")
    idx = torch.zeros(yd.shape, dtype=torch.uint8, device="cuda")
    a.check(a.lib.aocr_conv2d_forward(stream(), compute, a.ptr(xd), a.ptr(wd), a.ptr(bd), a.ptr(yd), a.ptr(idx), B, H, W, Cin, Cout,
                                      ks, pad, relu, pool))
    report(f"conv fwd c{compute} {Cin}->{Cout} k{ks} pool{pool}", yd, nhwc(y.detach()), tol)
    # backward: gradient at the (pooled) output.  Each kernel is checked on its own: un-pool against autograd
    # (a near-tie inside a window may pick another arg-max in fp32 than in fp64, so a handful of flipped windows
    # is tolerated), then dgrad / wgrad against torch.nn.grad on the DEVICE's dy.
    g = rnd(*y.shape, seed=4)
    y0.retain_grad()
    y.backward(g)
    Ho, Wo = y0.shape[2], y0.shape[3]
    gd = dev(nhwc(g))
    if pool:
        dy = torch.zeros(B, Ho, Wo, Cout, device="cuda")
        a.check(a.lib.aocr_unpool_relu_backward(stream(), a.ptr(gd), a.ptr(yd), a.ptr(idx), a.ptr(dy), B, Ho, Wo, Cout, pool))
        bad = ((dy.double().cpu() - nhwc(y0.grad)).abs() > 1e-6).sum().item()
        print(f"[parity] unpool pool{pool}: {bad} of {dy.numel()} elements differ (arg-max near-ties)")
        assert bad <= (max(4, dy.numel() // 20000) if compute == 0 else dy.numel() // 50)   # bf16 rounding flips more near-ties
    else:
        dy = gd
    dyr = dy.double().cpu().permute(0, 3, 1, 2).contiguous()
    dx_ref = torch.nn.grad.conv2d_input(x.shape, w.detach(), dyr, padding=pad)
    dw_ref = torch.nn.grad.conv2d_weight(x.detach(), w.shape, dyr, padding=pad)
    db_ref = dyr.sum(dim=(0, 2, 3))
    dx = torch.zeros(B, H, W, Cin, device="cuda")
    a.check(a.lib.aocr_conv2d_backward_data(stream(), compute, a.ptr(dy), a.ptr(wd), a.ptr(dx), B, H, W, Cin, Cout, ks, pad))
    report(f"conv dgrad c{compute} {Cin}->{Cout} k{ks} pool{pool}", dx, nhwc(dx_ref), tol * 3)
    dw = torch.zeros(Cout, ks, ks, Cin, device="cuda"); db = torch.zeros(Cout, device="cuda")
    a.check(a.lib.aocr_conv2d_backward_filter(stream(), compute, a.ptr(xd), a.ptr(dy), a.ptr(dw), a.ptr(db), B, H, W, Cin, Cout, ks, pad))
    scale = max(1.0, dw_ref.abs().max().item())
    report(f"conv wgrad c{compute} {Cin}->{Cout} k{ks} pool{pool}", dw / scale, dw_ref.permute(0, 2, 3, 1) / scale, tol * 3)
    report(f"conv bgrad c{compute}", db / scale, db_ref / scale, tol * 3)


@pytest.mark.parametrize("B,W", [(2, 36), (3, 100)])
def test_conv1(cuda, B, W):
    a = _lib()
    H = 32
    g0 = torch.Generator().manual_seed(5)
    img = torch.floor(torch.rand(B, 1, H, W, generator=g0, dtype=torch.float64) * 256)
    w = (rnd(64, 1, 3, 3, seed=2) / 3).requires_grad_(True); b = rnd(64, seed=3, scale=0.3).requires_grad_(True)
    y = F.max_pool2d(F.relu(F.conv2d((img - 128.0) / 128.0, w, b, padding=1)), 2, 2)
    xd = dev(img.reshape(B, H, W)); wd = dev(w.detach().reshape(64, 9)); bd = dev(b.detach())
    yd = torch.zeros(B, H // 2, W // 2, 64, device="cuda")
    a.check(a.lib.aocr_conv1_forward(stream(), a.ptr(xd), a.ptr(wd), a.ptr(bd), a.ptr(yd), B, H, W))
    report("conv1 fwd", yd, nhwc(y.detach()), 2e-5)
    g = rnd(*y.shape, seed=6)
    y.backward(g)
    dw = torch.zeros(64, 9, device="cuda"); db = torch.zeros(64, device="cuda")
    a.check(a.lib.aocr_conv1_backward(stream(), a.ptr(xd), a.ptr(wd), a.ptr(bd), a.ptr(dev(nhwc(g))), a.ptr(dw), a.ptr(db), B, H, W))
    sc = max(1.0, w.grad.abs().max().item())
    report("conv1 wgrad", dw / sc, w.grad.reshape(64, 9) / sc, 1e-4)
    report("conv1 bgrad", db / sc, b.grad / sc, 1e-4)


@pytest.mark.parametrize("rows,C,tb", [(150, 256, 0), (2 * 24, 512, 2), (5000, 512, 0)])
def test_batchnorm_relu(cuda, rows, C, tb):
    a = _lib()
    x = (rnd(rows, C, seed=1) * 2 + 0.3).requires_grad_(True)
    w = (rnd(C, seed=2).abs() + 0.1).requires_grad_(True); b = rnd(C, seed=3, scale=0.2).requires_grad_(True)
    rm = torch.zeros(C, dtype=torch.float64); rv = torch.ones(C, dtype=torch.float64)
    y = F.relu(F.batch_norm(x, rm, rv, w, b, training=True, momentum=0.1, eps=1e-5))
    xd = dev(x.detach()); yd = torch.zeros(rows, C, device="cuda")
    rmd = torch.zeros(C, device="cuda"); rvd = torch.ones(C, device="cuda")
    save = torch.zeros(2 * C, device="cuda"); scratch = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda")
    a.check(a.lib.aocr_batchnorm_relu_forward(stream(), a.ptr(xd), a.ptr(yd), a.ptr(dev(w.detach())), a.ptr(dev(b.detach())), a.ptr(rmd),
                                              a.ptr(rvd), a.ptr(save), a.ptr(scratch), rows, C, 1, 1, tb))
    yref = y.detach()
    if tb:
        yref = yref.reshape(tb, rows // tb, C).transpose(0, 1).reshape(rows, C)
    report(f"bn fwd rows={rows} C={C} tb={tb}", yd, yref, 2e-5)
    report("bn running_mean", rmd, rm, 1e-6); report("bn running_var", rvd, rv, 1e-5)
    g = rnd(rows, C, seed=7)
    y.backward(g)
    gd = g
    if tb:
        gd = g.reshape(tb, rows // tb, C).transpose(0, 1).reshape(rows, C)
    dx = torch.zeros(rows, C, device="cuda"); dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    a.check(a.lib.aocr_batchnorm_relu_backward(stream(), a.ptr(xd), a.ptr(yd), a.ptr(dev(gd)), a.ptr(dev(w.detach())), a.ptr(save), a.ptr(dx),
                                               a.ptr(dw), a.ptr(db), a.ptr(scratch), rows, C, tb))
    report("bn dx", dx, x.grad, 5e-5)
    sc = max(1.0, w.grad.abs().max().item())
    report("bn dw", dw / sc, w.grad / sc, 5e-5); report("bn db", db / sc, b.grad / sc, 5e-5)
    # eval mode
    a.check(a.lib.aocr_batchnorm_relu_forward(stream(), a.ptr(xd), a.ptr(yd), a.ptr(dev(w.detach())), a.ptr(dev(b.detach())), a.ptr(rmd),
                                              a.ptr(rvd), a.ptr(save), a.ptr(scratch), rows, C, 0, 0, 0))
    ye = F.relu(F.batch_norm(x.detach(), rm, rv, w.detach(), b.detach(), training=False, eps=1e-5))
    report("bn eval", yd, ye, 2e-5)


@pytest.mark.parametrize("B,inp,H", [(5, 512, 32), (64, 512, 256), (33, 64, 64)])
@pytest.mark.parametrize("compute", [0, 1])
def test_lstm_cell(cuda, B, inp, H, compute):
    import oracle_torch as O
    a = _lib()
    x = rnd(B, inp, seed=1); h = rnd(B, H, seed=2); c = rnd(B, H, seed=3)
    Wi = rnd(4 * H, inp, seed=4) / inp ** 0.5; bi = rnd(4 * H, seed=5, scale=0.1)
    Wh = rnd(4 * H, H, seed=6) / H ** 0.5; bh = rnd(4 * H, seed=7, scale=0.1)
    c2, h2, cache = O.lstm_cell_fwd(x, c, h, Wi, bi, Wh, bh)
    cd = torch.zeros(B, H, device="cuda"); hd = torch.zeros(B, H, device="cuda"); gd = torch.zeros(B, 4 * H, device="cuda")
    cpd = dev(c)
    a.check(a.lib.aocr_lstm_cell_forward(stream(), compute, a.ptr(dev(x)), inp, a.ptr(dev(h)), a.ptr(cpd), a.ptr(dev(Wi)), a.ptr(dev(bi)),
                                         a.ptr(dev(Wh)), a.ptr(dev(bh)), a.ptr(cd), a.ptr(hd), a.ptr(gd), B, H))
    tol = 2e-5 if compute == 0 else 3e-2
    report(f"lstm c c{compute}", cd, c2, tol); report(f"lstm h c{compute}", hd, h2, tol)
    report(f"lstm gates c{compute}", gd, torch.cat(cache[:4], 1), tol)
    if compute == 0:
        dc = rnd(B, H, seed=8); dh = rnd(B, H, seed=9)
        dx, dcp, dhp, dz = O.lstm_cell_bwd(dc, dh, cache, x, c, h, Wi, Wh)
        dzd = torch.zeros(B, 4 * H, device="cuda"); dcpd = torch.zeros(B, H, device="cuda")
        a.check(a.lib.aocr_lstm_cell_backward(stream(), a.ptr(dev(dc)), a.ptr(dev(dh)), a.ptr(gd), a.ptr(cpd), a.ptr(cd), a.ptr(dzd), a.ptr(dcpd), B, H))
        report("lstm dz", dzd, dz, 5e-5); report("lstm dc_prev", dcpd, dcp, 5e-5)


@pytest.mark.parametrize("B,T,Hd", [(3, 8, 32), (64, 24, 512), (5, 199, 512)])
def test_attention(cuda, B, T, Hd):
    a = _lib()
    ctx = rnd(B, T, Hd, seed=1); q = rnd(B, Hd, seed=2) * 0.3
    s = torch.bmm(ctx, q.unsqueeze(2)).squeeze(2); al = torch.softmax(s, 1); c = torch.bmm(al.unsqueeze(1), ctx).squeeze(1)
    ad = torch.zeros(B, T, device="cuda"); cd = torch.zeros(B, 2 * Hd, device="cuda")
    ctxd = dev(ctx); qd = dev(q)
    a.check(a.lib.aocr_attention_forward(stream(), a.ptr(ctxd), a.ptr(qd), a.ptr(ad), a.ptr(cd), 2 * Hd, B, T, Hd))
    report("attn a", ad, al, 2e-5); report("attn c", cd[:, :Hd], c, 5e-5)
    dc = rnd(B, Hd, seed=3)
    da = torch.bmm(ctx, dc.unsqueeze(2)).squeeze(2); ds = al * (da - (al * da).sum(1, keepdim=True)); dq = torch.bmm(ds.unsqueeze(1), ctx).squeeze(1)
    dsd = torch.zeros(B, T, device="cuda"); dqd = torch.zeros(B, Hd, device="cuda")
    dcd = torch.zeros(B, 2 * Hd, device="cuda"); dcd[:, :Hd] = dev(dc)
    a.check(a.lib.aocr_attention_backward(stream(), a.ptr(ctxd), a.ptr(qd), a.ptr(dev(al)), a.ptr(dcd), 2 * Hd, a.ptr(dsd), a.ptr(dqd), B, T, Hd))
    report("attn ds", dsd, ds, 5e-5); report("attn dq", dqd, dq, 1e-4)


def test_logsoftmax_nll(cuda):
    a = _lib()
    rows, V, ld = 333, 39, 40
    x = rnd(rows, V, seed=1) * 3
    g0 = torch.Generator().manual_seed(2)
    y = torch.randint(1, V + 1, (rows,), generator=g0)
    y[::7] = 1                                           # PAD rows carry weight 0
    lp = torch.log_softmax(x, 1)
    w = torch.ones(V, dtype=torch.float64); w[0] = 0
    nll = -(w[y - 1] * lp[torch.arange(rows), y - 1])
    scale = 1.0 / 64
    dl = scale * w[y - 1].unsqueeze(1) * (torch.exp(lp) - F.one_hot(y - 1, V))
    xd = torch.zeros(rows, ld, device="cuda"); xd[:, :V] = dev(x)
    lpd = torch.zeros(rows, V, device="cuda"); dld = torch.ones(rows, ld, device="cuda"); nd = torch.zeros(rows, device="cuda")
    a.check(a.lib.aocr_logsoftmax_nll(stream(), a.ptr(xd), ld, a.ptr(dev(y, torch.int32)), a.ptr(lpd), a.ptr(dld), a.ptr(nd), rows, V, scale))
    report("logp", lpd, lp, 1e-5); report("nll", nd, nll, 1e-5); report("dlogits", dld[:, :V], dl, 1e-6)
    assert float(dld[:, V:].abs().max()) == 0.0


@pytest.mark.parametrize("kin,kout", [(1, 1), (1, 5), (5, 5), (3, 3)])
def test_beam_select(cuda, kin, kout):
    import oracle_torch as O
    a = _lib()
    B, V = 7, 39
    lp = torch.log_softmax(rnd(B * kin, V, seed=kin * 10 + kout) * 3, 1)
    prev = torch.randint(1, 8, (B * kin,), generator=torch.Generator().manual_seed(3)).to(torch.int32)
    bs = rnd(B, kin, seed=4)
    if kin == 1:
        vals, raw = O._topk_sorted(lp, kout); toks = raw + 1; par = torch.zeros_like(raw)
    else:
        l2 = lp.clone(); fin = (prev == 1) | (prev == 3); l2[fin, 0] = 0.0
        tot = (l2.view(B, kin, V) + bs.unsqueeze(2)).reshape(B, kin * V)
        vals, raw = O._topk_sorted(tot, kout); toks = raw % V + 1; par = raw // V
    bsd = torch.zeros(B, max(kin, kout), device="cuda").reshape(-1)
    bsd[:B * kin] = dev(bs).reshape(-1)
    tk = torch.zeros(B, kout, dtype=torch.int32, device="cuda"); pr = torch.zeros(B, kout, dtype=torch.int32, device="cuda")
    a.check(a.lib.aocr_beam_select(stream(), a.ptr(dev(lp)), a.ptr(dev(prev, torch.int32)) if kin > 1 else None, a.ptr(bsd), a.ptr(tk), a.ptr(pr), B, kin, kout, V))
    assert torch.equal(tk.cpu().long(), toks), (tk.cpu(), toks)
    assert torch.equal(pr.cpu().long(), par)
    report("beam scores", bsd[:B * kout].reshape(B, kout), vals, 1e-5)


# ------------------------------------------------------------------------------------------------ round 3: the rest of the module surface
@pytest.mark.parametrize("flag,fn", [(2, torch.relu), (4, torch.tanh)])
def test_gemm_activation_flags(cuda, flag, fn):
    """aocr_gemm's flag bits: 2 = ReLU, 4 = tanh on the result (nn.Tanh behind nn.LinearNoBias, LSTM.lua:155-157); 1 = accumulate."""
    a = _lib()
    M, N, K = 70, 96, 128
    A = rnd(M, K, seed=1); Bm = rnd(N, K, seed=2)
    Cd = torch.zeros(M, N, device="cuda")
    a.check(a.lib.aocr_gemm(stream(), 0, a.ptr(dev(A)), K, 1, a.ptr(dev(Bm)), K, 1, a.ptr(Cd), N, M, N, K, None, flag))
    report(f"gemm flag {flag}", Cd, fn(A @ Bm.t()), 2e-4)
    a.check(a.lib.aocr_gemm(stream(), 0, a.ptr(dev(A)), K, 1, a.ptr(dev(Bm)), K, 1, a.ptr(Cd), N, M, N, K, None, 1))
    report("gemm accumulate on top", Cd, fn(A @ Bm.t()) + A @ Bm.t(), 4e-4)
    assert a.lib.aocr_gemm(stream(), 0, a.ptr(dev(A)), K, 1, a.ptr(dev(Bm)), K, 1, a.ptr(Cd), N, M, N, K, None, 5) != 0     # activation on an accumulating product: refused


@pytest.mark.parametrize("n", [5, 4096, 100003])
def test_pointwise(cuda, n):
    a = _lib()
    x = rnd(n, seed=1); y = rnd(n, seed=2)
    out = torch.zeros(n, device="cuda")
    for op, ref in ((0, x + y), (1, x * (1 - y * y)), (2, x * (y > 0)), (3, torch.relu(x))):
        a.check(a.lib.aocr_pointwise(stream(), op, a.ptr(dev(x)), a.ptr(dev(y)) if op != 3 else None, a.ptr(out), n))
        report(f"pointwise op {op} n={n}", out, ref, 1e-6)
    buf = dev(x).clone()                                              # in place, from an address that is not 16-byte aligned
    a.check(a.lib.aocr_pointwise(stream(), 0, C.c_void_p(buf.data_ptr() + 4), C.c_void_p(buf.data_ptr() + 4), C.c_void_p(buf.data_ptr() + 4), n - 1))
    report("pointwise in place, unaligned", buf[1:], 2 * x[1:], 1e-6)


def test_lookup_forward_backward(cuda):
    """nn.LookupTable (LSTM.lua:55-56): gather of 1-based ids; accGradParameters = scatter-add of the output gradient rows."""
    a = _lib()
    V, E, n = 39, 20, 777
    W = rnd(V, E, seed=1)
    ids = torch.randint(1, V + 1, (n,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    out = torch.zeros(n, E, device="cuda")
    a.check(a.lib.aocr_lookup_forward(stream(), a.ptr(dev(W)), a.ptr(dev(ids, torch.int32)), a.ptr(out), n, E))
    report("lookup forward", out, W[ids.long() - 1].float(), 0.0)
    g = rnd(n, E, seed=3)
    dW = torch.zeros(V, E, device="cuda")
    a.check(a.lib.aocr_lookup_backward(stream(), a.ptr(dev(g)), a.ptr(dev(ids, torch.int32)), a.ptr(dW), n, E, V))
    ref = torch.zeros(V, E, dtype=torch.float64).index_add_(0, ids.long() - 1, g)
    report("lookup backward", dW, ref, 1e-4)


@pytest.mark.parametrize("B,H,inp", [(5, 32, 52), (33, 64, 532), (7, 16, 20)])
def test_lstm_cell_forward_zx(cuda, B, H, inp):
    """The cell with the input part pre-computed (input widths that are not multiples of 16: the decoder's first layer sees E + Hd
    columns): zx = W_i2h x + b_i2h + b_h2h by aocr_gemm with the summed bias, then aocr_lstm_cell_forward_zx; against LSTM.lua:79-105."""
    a = _lib()
    x = rnd(B, inp, seed=1); hp = rnd(B, H, seed=2); cp = rnd(B, H, seed=3)
    Wi = rnd(4 * H, inp, seed=4, scale=0.3); bi = rnd(4 * H, seed=5); Wh = rnd(4 * H, H, seed=6, scale=0.3); bh = rnd(4 * H, seed=7)
    z = x @ Wi.t() + bi + hp @ Wh.t() + bh
    i, f, o, g = (torch.sigmoid(z[:, k * H:(k + 1) * H]) if k < 3 else torch.tanh(z[:, k * H:(k + 1) * H]) for k in range(4))
    c_ref = f * cp + i * g; h_ref = o * torch.tanh(c_ref)
    bsum = torch.zeros(4 * H, device="cuda"); zx = torch.zeros(B, 4 * H, device="cuda")
    a.check(a.lib.aocr_pointwise(stream(), 0, a.ptr(dev(bi)), a.ptr(dev(bh)), a.ptr(bsum), 4 * H))
    a.check(a.lib.aocr_gemm(stream(), 0, a.ptr(dev(x)), inp, 1, a.ptr(dev(Wi)), inp, 1, a.ptr(zx), 4 * H, B, 4 * H, inp, a.ptr(bsum), 0))
    c = torch.zeros(B, H, device="cuda"); h = torch.zeros(B, H, device="cuda"); gates = torch.zeros(B, 4 * H, device="cuda")
    a.check(a.lib.aocr_lstm_cell_forward_zx(stream(), 0, a.ptr(zx), 4 * H, a.ptr(dev(hp)), a.ptr(dev(cp)), a.ptr(dev(Wh)), a.ptr(c), a.ptr(h), a.ptr(gates), B, H))
    report(f"lstm cell zx c B={B} H={H} in={inp}", c, c_ref, 2e-5)
    report(f"lstm cell zx h B={B} H={H} in={inp}", h, h_ref, 2e-5)
    report("lstm cell zx gates", gates, torch.cat([i, f, o, g], 1), 2e-5)
