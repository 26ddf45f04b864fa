"""Debugging aid: cycles per phase of the two-chain decoder kernels (dec_chain.hip), workgroup 0, summed over the L steps.
Needs the stamps build:  make -C torch-attention-ocr_amd/csrc stamps;  AOCR_LIB=.../libaocr_stamps.so python tools/ch_stamp.py [B]"""
import os, sys
os.environ["AOCR_DC_STAMPS"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("tests", "oracle", "torch-attention-ocr_amd"):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch
from test_step_gpu import make
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, L = 256, 24
m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=B, W=W, maxlen=L - 1, compute="bf16", max_decoder_l=L, max_beam=1)
for _ in range(3):
    m.train_forward_backward(batch)
torch.cuda.synchronize()
for tap, names in (("dc_stamps", ["P1<0>", "P1<1>", "P2<0>", "P2<1>", "P3<0>", "P3<1>", "P4<0>", "P4<1>"]),
                   ("dc_bstamps", ["B1<0>", "B1<1>", "B2<0>", "B2<1>", "B3<0>", "B3<1>", "B4<0>", "B4<1>", "B5<0>", "B5<1>", "B6<0>", "B6<1>"])):
    s = m.get_tensor(tap).view(torch.int64).cpu().tolist()
    tot = sum(s[:len(names)])
    if tot == 0:
        continue
    print(tap, "workgroup 0 (member 0 of group 0), cycles per step (100 MHz ticks x ? -- s_memtime counts shader-clock-independent ticks)")
    for n, v in zip(names, s):
        print(f"  {n:8s} {v / L:9.0f}  {100.0 * v / tot:5.1f} %")
    print(f"  total {tot / L:9.0f} per step;  aux {s[12:]}")
m.shutdown()
