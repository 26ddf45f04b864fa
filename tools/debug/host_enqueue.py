"""Host time to ENQUEUE a training step against the time the GPU takes to run it (is the host ever the bound?): python tools/debug/host_enqueue.py [c3|ref|c2]"""
import sys, time
for d in ("tests", "oracle", "torch-attention-ocr_amd"):
    sys.path.insert(0, d)
import torch
from test_step_gpu import make
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfgs = {"c3": (dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), 256, 256, "bf16", 24),
        "ref": (dict(enc_hidden=512, enc_layers=1, dec_layers=2, input_feed=True), 400, 100, "bf16", 24),
        "c2": (dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), 64, 100, "f32", 24)}
cfg, B, W, comp, L = cfgs[w]
m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=L - 1, compute=comp, max_decoder_l=L, max_beam=1)
images, targets, targets_eval = m._upload(batch)
for _ in range(5):
    m.train_step_device(images, targets, targets_eval, B)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for _ in range(N):
    m.train_step_device(images, targets, targets_eval, B)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{w}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms per step; GPU {1e3 * (t2 - t0) / N:.3f} ms per step")
