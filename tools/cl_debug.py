import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("torch-attention-ocr_amd", "oracle", "tests"): sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
from test_step_gpu import make, relerr
He, B, W, Le = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 16, 40, 1))]
cfg = dict(enc_hidden=He, enc_layers=Le, dec_layers=2, input_feed=True)
out = {}
for name, env in (("cluster", {}), ("step", {"AOCR_NO_SEQ": "1"})):
    os.environ.pop("AOCR_NO_SEQ", None)
    os.environ.update(env)
    m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=6, compute="bf16")
    loss = m.train_forward_backward(batch)
    out[name] = dict(loss=loss, dfeats=m.get_tensor("dfeats").clone(), dz0=m.get_tensor("enc_dz0").clone(), cs0=m.get_tensor("enc_cs0").clone(), g0=m.get_tensor("enc_gates0").clone(), dz1=m.get_tensor("enc_dz1").clone(), grads={k: v.clone() for k, v in m.get_gradients().items()})
    if name == "cluster": print("cl_err", m.get_tensor("cl_err").view(torch.int32).tolist())
    m.shutdown()
a, b = out["step"], out["cluster"]
print("loss", a["loss"], b["loss"], "dfeats rel", relerr(b["dfeats"], a["dfeats"]))
df = (b["dfeats"] - a["dfeats"]).abs()          # (T,B,512)
print("dfeats err by t:", [f"{df[t].max().item():.2e}" for t in range(df.shape[0])])
print("dfeats err by b:", [f"{df[:, r].max().item():.2e}" for r in range(df.shape[1])])
for k in a["grads"]:
    if k.startswith("enc_"):
        print(f"{k:20s} rel {relerr(b['grads'][k], a['grads'][k]):.3e}")

for nm in ("dz0", "dz1"):
    e = (b[nm] - a[nm]).abs()
    print(nm, "max err", e.max().item(), "ref max", a[nm].abs().max().item())
    idx = (e > 1e-3 * a[nm].abs().max()).nonzero()
    print("  bad entries", idx.shape[0], "of", e.numel())
    if idx.shape[0]:
        print("  t values", sorted(set(idx[:, 0].tolist())))
        print("  rows", sorted(set(idx[:, 1].tolist())))
        cols = sorted(set(idx[:, 2].tolist())); print("  cols", cols[:40], "...", len(cols))

e = (b["cs0"] - a["cs0"]).abs(); print("cs0 max err", e.max().item(), "rows bad", sorted(set((e > 1e-4).nonzero()[:, 1].tolist())))
T, Bb, H4 = a["g0"].shape; H = H4 // 4
gc = b["g0"].reshape(T, Bb, H, 4).permute(0, 1, 3, 2).reshape(T, Bb, H4)      # cluster layout [T][B][He][4] -> planes
e = (gc - a["g0"]).abs(); print("gates max err", e.max().item(), "rows bad", sorted(set((e > 1e-4).nonzero()[:, 1].tolist())), "t bad", sorted(set((e > 1e-4).nonzero()[:, 0].tolist())))
