"""Where a 256 x 256 halo-kernel workgroup of conv3 / conv4 / conv5 / conv6 forward spends its time at the C3 shape (the TAG = 1 instantiation's wall-clock probes):
AOCR_PROBE=1 AOCR_PROBE_LAYER=3|4|5 python tools/debug/conv_probe.py   (no AOCR_PROBE_LAYER: conv6)"""
import os, sys
for d in ("tests", "oracle", "torch-attention-ocr_amd"):
    sys.path.insert(0, d)
import torch
from test_step_gpu import make
cfg = dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
m, O, ocfg, P, st, batch = make(cfg, B=256, W=256, maxlen=11, compute="bf16", max_decoder_l=12, max_beam=1)
for _ in range(2):
    m.train_forward_backward(batch)
torch.cuda.synchronize()
ms, fl = m.profile_kernel(0, 20)
print(f"layer {os.environ.get('AOCR_PROBE_LAYER', '6')}: {ms * 1e3:.1f} us per launch")
