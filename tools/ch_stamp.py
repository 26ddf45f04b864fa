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
s = m.get_tensor("dc_stamps").view(torch.int64).cpu().tolist()
r = m.get_tensor("dc_times").view(torch.int32).cpu().tolist()
names = ["P1<0>", "P1<1>", "P2<0>", "P2<1>", "P3<0>", "P3<1>", "P4<0>", "P4<1>"]
tot = sum(s)
print("forward, workgroup 0 (member 0 of group 0), shader cycles per step: wait for the operand (land) + the rest of the phase; poll retries per step")
for k, n in enumerate(names):
    print(f"  {n:6s} land {s[8 + k] / L:7.0f}  rest {s[k] / L:7.0f}  retries {r[k] / L:5.2f}")
print(f"  total {tot / L:9.0f} per step")
b = m.get_tensor("dc_bstamps").view(torch.int64).cpu().tolist()
bn = ["B2<0>", "B2<1>", "B3<0>", "B3<1>", "B4<0>", "B4<1>", "B5<0>", "B5<1>", "B6<0>", "B6<1>"]
print("backward, workgroup 0, shader cycles per step")
for k, n in enumerate(bn):
    print(f"  {n:6s} {b[k] / L:8.0f}" + (f"   + operand wait {b[10 + k - 4] / L:7.0f}" if k >= 4 else ""))
print(f"  total {sum(b[:16]) / L:9.0f} per step; d z poll retries per step (wave 0 of workgroup 0) {m.get_tensor('dc_times').view(torch.int32)[8].item() / L:.2f}")
m.shutdown()
