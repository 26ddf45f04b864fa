"""Cycles per phase of the beam-search chain kernel (stamps build): AOCR_LIB=.../libaocr_stamps.so python tools/debug/beam_stamp.py [beam]"""
import os, sys
os.environ["AOCR_DC_STAMPS"] = "beam" if len(sys.argv) > 1 and int(sys.argv[1]) > 1 else "1"
sys.path.insert(0, "torch-attention-ocr_amd")
import torch, aocr, aocr.synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, W, L = 256, 256, 50
dev = torch.device("cuda:0")
m = aocr.Model().create(dict(encoder_num_hidden=256, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=B, max_img_w=W,
                             max_decoder_l=L, max_beam=k, compute="bf16", learning_rate=0.1, seed=910820))
img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=1234, max_len=L - 1, H=32)
images = torch.from_numpy(img).to(device=dev, dtype=torch.float32)
targets = torch.from_numpy(tgt).to(dev); targets_eval = torch.from_numpy(tge).to(dev)
for _ in range(3):
    m.decode_device(images, targets, targets_eval, k)
torch.cuda.synchronize()
s = m.get_tensor("dc_stamps").view(torch.int64).cpu().tolist()
r = m.get_tensor("dc_times").view(torch.int32).cpu().tolist()
names = ["P1<0>", "P1<1>", "P2<0>", "P2<1>", "P3<0>", "P3<1>", "P4<0>", "P4<1>"]
print(f"beam {k}: workgroup 0 (member 0 of group 0 = an image owner of chain 0), shader cycles per step; land of P1<0> includes the pre-fill and P5")
for i, n in enumerate(names):
    print(f"  {n:6s} land {s[8 + i] / L:7.0f}  rest {s[i] / L:7.0f}  retries {r[i] / L:5.2f}")
print(f"  total {sum(s) / L:9.0f} per step")
print(f"  prologue {r[9] * 16} cycles, whole kernel {r[10] * 16} cycles = {r[11] / 100:.1f} us (100 MHz counter) -> {r[10] * 16 / (r[11] / 100) / 1e3:.2f} GHz")
print("  P5 of the owner, cycles per step: poll+partial sums %d, barrier %d, softmax+reset %d, barrier %d, k-best %d, publish %d" % tuple(r[12 + i] * 16 // L for i in range(6)))
import numpy as np
a = np.array(r[32:32 + 256]); b = np.array(r[288:288 + 256]); c = np.array(r[544:544 + 256])
t0 = a[a > 0].min() if (a > 0).any() else 0
print("  per workgroup of the last launch (100 MHz ticks from the first start): start / loop start / end; workgroup = xcd + 8 * member (+ 256 * ...)")
for w in list(range(0, 256, 37)) + [1, 2, 8, 9, 10, 255]:
    print(f"    wid {w:3d} xcd {w & 7} member {w >> 3}: start {(a[w] - t0) / 100:8.1f} us  loop {(b[w] - t0) / 100:8.1f}  end {(c[w] - t0) / 100:8.1f}")
act = c > 0
print(f"  starts: min {(a[act].min() - t0) / 100:.1f} max {(a[act].max() - t0) / 100:.1f} us; loop starts min {(b[act].min() - t0) / 100:.1f} max {(b[act].max() - t0) / 100:.1f}; ends min {(c[act].min() - t0) / 100:.1f} max {(c[act].max() - t0) / 100:.1f}")
