"""Three optimisation steps twice (optionally with a switch the second time): relative gradient difference per step.  python tools/debug/side_cmp.py [ENV=1 ...]"""
import os, sys
for d in ("tests", "oracle", "torch-attention-ocr_amd"):
    sys.path.insert(0, d)
import torch
from test_step_gpu import make
cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
def run(env):
    for k, v in env.items(): os.environ[k] = v
    m, O, ocfg, P, st, batch = make(cfg, B=32, W=256, maxlen=11, compute="bf16", max_decoder_l=12, max_beam=1)
    images, targets, targets_eval = m._upload(batch)
    grads = []
    for i in range(3):
        m.train_step_device(images, targets, targets_eval, 32)
        grads.append(m.grad_params.clone())
    torch.cuda.synchronize()
    assert m.cluster_status() == 0
    names = m.get_gradients().keys()
    lay = {k: v.clone() for k, v in m.get_gradients().items()}
    m.shutdown()
    for k in env: del os.environ[k]
    return grads, lay
env = dict(a.split("=") for a in sys.argv[1:])
g0, l0 = run({}); g1, l1 = run({}); g2, l2 = run(env)
for i in range(3):
    print(f"step {i}: same settings twice {((g0[i] - g1[i]).norm() / g0[i].norm()).item():.3e}   with {env}: {((g0[i] - g2[i]).norm() / g0[i].norm()).item():.3e}")
for k in l0:
    e = ((l0[k] - l2[k]).norm() / (l0[k].norm() + 1e-30)).item()
    if e > 1e-2: print(f"   last step {k}: {e:.3e}")
