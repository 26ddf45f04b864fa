import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for d in ("tests", "oracle", "torch-attention-ocr_amd"):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch
from test_step_gpu import make, relerr
B, W, maxlen = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
out = {}
for knob in (os.environ.get("KA", "0"), "1"):
    os.environ["AOCR_NO_DEC_CHAINS"] = knob
    m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=maxlen, compute="bf16", max_decoder_l=maxlen + 1, max_beam=1)
    for rep in range(2):
        loss = m.train_forward_backward(batch)
    out[knob if knob not in out else knob + "b"] = dict(grads={k: v.clone() for k, v in m.get_gradients().items()}, taps={k: m.get_tensor(k).clone() for k in ("ds_all", "dq_all", "dpre_all", "dcat_all", "dcontext", "enc_dz0", "dh_rec0", "dh_rec1", "dc_st0", "dc_st1", "dfeed0")})
    print("cl_err", m.get_tensor("cl_err").view(torch.int32)[:6].tolist())
    m.shutdown()
a = out["1b"] if "1b" in out else out["1"]
b = out[os.environ.get("KA", "0")]
for k in a["taps"]:
    x, y = a["taps"][k], b["taps"][k]
    if k == "dcat_all": x, y = x[..., :512], y[..., :512]
    print(f"tap {k:10s} equal={torch.equal(x, y)} rel={relerr(y, x):.3e}")
    if not torch.equal(x, y) and x.dim() == 2:
        bad = (x != y).nonzero()
        print("   rows", sorted(set(bad[:, 0].tolist())), "cols", sorted(set(bad[:, 1].tolist()))[:40], "n", len(bad))
for k in a["grads"]:
    r = relerr(b["grads"][k], a["grads"][k])
    if r > 1e-6: print(f"grad {k:24s} rel={r:.3e}")
