"""Cycles per phase of the encoder cluster forward kernel, workgroup 0 (stamps build): AOCR_LIB=.../libaocr_stamps.so python tools/debug/enc_stamp.py [workload c3|c5]"""
import os, sys
for d in ("tests", "oracle", "torch-attention-ocr_amd"):
    sys.path.insert(0, d)
import torch
from test_step_gpu import make
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
if w == "c3": cfg, B, W, H = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), 256, 256, 32
else: cfg, B, W, H = dict(enc_hidden=512, enc_layers=1, dec_layers=2, input_feed=True), 16, 1024, 32
m, O, ocfg, P, st, batch = make(cfg, B=B, W=W, maxlen=11, compute="bf16", max_decoder_l=12, max_beam=1)
for _ in range(3):
    m.train_forward_backward(batch)
torch.cuda.synchronize()
T = (H // 16 - 1) * (W // 4 - 1)
r = m.get_tensor("dc_times").view(torch.int32).cpu().tolist()
names = ["loop top", "issue polls + output stores", "wait for the polls", "tag check / spin", "payload to LDS + barrier", "MFMA", "zx + cell + publish + next zx DMA"]
tot = 0
for k, n in enumerate(names):
    c = r[1000 + k] * 16 / T; tot += c
    print(f"  {n:36s} {c:8.0f} cycles / step")
print(f"  (of the second line: issuing the polls {r[1007] * 16 / T:.0f})")
print(f"  total {tot + r[1007] * 16 / T:.0f} cycles / step over T = {T}")
if any(r[1010:1018]):
    print("  per output store (cs, hsb, ctx, gates 0..3):", [int(r[1010 + k] * 16 / T) for k in range(7)])
bn = ["barrier at the loop end + loop top", "prefetch DMA issue + MFMA", "partial tiles out (granule stores / LDS)", "barrier", "polls issued + d z stores", "wait for the polls + epilogue inputs from LDS", "tag check / spin + partial sums", "cell backward + d z to LDS"]
print("backward kernel, workgroup 0:")
tb = 0
for k, n in enumerate(bn):
    c = r[1020 + k] * 16 / T; tb += c
    print(f"  {n:46s} {c:8.0f} cycles / step")
print(f"  total {tb:.0f} cycles / step")
