"""Diagnostic: where in the CNN backward pass does the bf16 path leave the bf16-operand oracle?  Compares the gradient maps after
BatchNorm-7 backward (stage 1) and after conv7 dgrad + un-pool (stage 2) with the oracle's retained gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-attention-ocr_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import torch.nn.functional as F
import oracle_torch as O
from test_step_gpu import make

saved = {}
orig = F.conv2d
def conv2d(x, w, b, stride=1, padding=0):
    y = orig(x, w, b, stride=stride, padding=padding)
    if y.requires_grad:
        y.retain_grad(); saved[w.shape] = y
    return y
O.F.conv2d = conv2d

def cos(a, b):
    a = a.double().reshape(-1); b = b.double().reshape(-1); return float(a @ b / (a.norm() * b.norm() + 1e-300))

cfgkw = dict(enc_hidden=64, enc_layers=1, dec_layers=2, input_feed=True)
B, W = 32, 128
for stage in (1, 2, 0):
    os.environ["AOCR_DBG_STOP"] = str(stage)
    m, _, ocfg, P, st, batch = make(cfgkw, B=B, W=W, maxlen=10, compute="bf16", max_decoder_l=12, max_beam=1)
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    with O.operand_rounding("bf16"):
        loss_q, Gq, rq, _ = O.train_step_autograd(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    if stage:
        g0 = m.get_tensor("g0")                       # (B, H, W, 512) channels-last
        key = (512, 512, 2, 2) if stage == 1 else (512, 512, 3, 3)
        ref = saved[key].grad.permute(0, 2, 3, 1)
        print(f"stage {stage}: g0 {tuple(g0.shape)} vs oracle {tuple(ref.shape)}: cos {cos(g0, ref):.6f} rel {(g0.double() - ref).abs().max().item() / ref.abs().max().item():.3e}")
        d = (g0.double() - ref).abs(); thr = 0.02 * ref.abs().max().item()
        print(f"   elements off by > 2% of max: {(d > thr).sum().item()} of {d.numel()}; oracle nonzero frac {(ref != 0).double().mean().item():.3f}, hip nonzero frac {(g0 != 0).double().mean().item():.3f}")
        print("   dfeats cos", cos(m.get_tensor("dfeats").transpose(0, 1), saved_dfeats) if 'saved_dfeats' in globals() else "n/a")
    else:
        grads = m.get_gradients()
        for k in ("cnn.bn7.w", "cnn.conv7.w", "cnn.conv6.w", "cnn.conv5.w", "cnn.conv2.w"):
            print(f"final {k}: cos {cos(grads[k], Gq[k]):.6f}")
    m.shutdown()
