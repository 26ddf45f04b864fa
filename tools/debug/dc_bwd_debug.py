"""Debugging aid: gradient-by-gradient comparison of the decoder cluster kernels against the launch chain."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "torch-attention-ocr_amd"))
import torch
from test_step_gpu import make, relerr
B, W, maxlen = 32, 72, 6
out = {}
for knob in ("0", "1"):
    os.environ.pop("AOCR_NO_DEC_CLUSTER", None)
    if knob == "1":
        os.environ["AOCR_NO_DEC_CLUSTER"] = "1"
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=B, W=W, maxlen=maxlen, compute="bf16", max_decoder_l=maxlen + 1, max_beam=1)
    loss = m.train_forward_backward(batch)
    print("cl_err", m.get_tensor("cl_err").view(torch.int32)[:4].tolist())
    taps = {k: m.get_tensor(k).clone() for k in ('ds_all', 'dq_all', 'dcat_all', 'dpre_all')}
    out[knob] = dict(taps=taps, loss=loss, grads={k: v.clone() for k, v in m.get_gradients().items()}, dctx=m.get_tensor("dcontext").clone())
    m.shutdown()
a, b = out["1"], out["0"]
for k in a["taps"]:
    x, y = a["taps"][k].double(), b["taps"][k].double()
    if k == "dcat_all": x, y = x[:, :, :512], y[:, :, :512]
    print(k, "rel", ((x - y).abs().max() / x.abs().max()).item(), "|ref|", x.abs().max().item(), "per-step rel", [round(((x[t] - y[t]).abs().max() / (x[t].abs().max() + 1e-30)).item(), 4) for t in range(x.shape[0])])
print("loss", a["loss"], b["loss"], "dctx rel", relerr(b["dctx"], a["dctx"]))
for k in a["grads"]:
    print(f"{k:20s} rel {relerr(b['grads'][k], a['grads'][k]):.3e}  |ref| {a['grads'][k].abs().max().item():.3e}")
