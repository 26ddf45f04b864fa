import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "torch-attention-ocr_amd"))
import numpy as np, torch
from test_step_gpu import make, relerr
cfg = dict(enc_hidden=128, enc_layers=1, dec_layers=2, input_feed=True)
for no_seq in ("0", "1"):
    os.environ["AOCR_NO_SEQ"] = no_seq
    m, O, ocfg, P, st, batch = make(cfg, B=16, W=36, maxlen=6, compute="bf16")
    img, tgt, tge = (torch.from_numpy(np.asarray(x)) for x in batch[:3])
    loss_ref, G, aux, _ = O.train_step_manual(P, st, ocfg, img, tgt, tge)
    loss = m.train_forward_backward(batch)
    lg = m.get_tensor("logits")[:, :, :ocfg.vocab]
    print("no_seq", no_seq, "logits max-abs vs oracle", (lg.double() - aux["logits"]).abs().max().item(), "context", (m.get_tensor("context").double() - aux["context"]).abs().max().item(), "loss", loss, float(loss_ref) * 16)
    m.shutdown()
