"""Debugging aid: cycles per phase of the decoder cluster kernels (workgroup 0, summed over the L steps) and a real-time timeline of one
step of group 0.  Needs a library built with the stamps compiled in:
    make -C torch-attention-ocr_amd/csrc FLAGS_dec_cluster=-DDC_DEBUG_STAMPS   (touch dec_cluster.hip first)
The stamps' global stores perturb the kernels (the compiler inserts vmcnt waits around them): trust the proportions, not the totals."""
import os, sys
os.environ["AOCR_DC_STAMPS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "torch-attention-ocr_amd"))
import torch
from test_step_gpu import make
B, W, L = 256, 256, 24
m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=B, W=W, maxlen=L - 1, compute="bf16", max_decoder_l=L, max_beam=1)
for _ in range(3):
    m.train_forward_backward(batch)
torch.cuda.synchronize()
s = m.get_tensor("dc_stamps").view(torch.int64).cpu().tolist()
names = ["gather out", "layer 1 rest", "gather h1", "layer 2", "gather h2", "attn publish", "gather c", "out", "(gathers: first poll round)", "(gathers: retries)", "attn scores", "attn softmax", "attn context", "(gathers: LDS write + barrier)", "-", "group in one XCD"]
tot = sum(s)
raw = {14, 15}
tot = sum(v for i, v in enumerate(s) if i not in {8, 9, 13, 14, 15})
for i, (n, v) in enumerate(zip(names, s)):
    if i in raw:
        print(f"{n:30s} {v:9d} (total over {L} steps)")
    else:
        print(f"{n:30s} {v / L:9.0f} cycles/step  {100.0 * v / tot:5.1f} %")
print("total cycles/step", tot / L)
tm = m.get_tensor("dc_times").view(torch.int64).cpu().view(32, 4, 16)
t0 = int(tm[:, 0, 0:4].min())
print("timeline of step 10, group 0 (ns from the first 'out' flag), min .. max over the 32 members (and the 4 waves where per wave)")
for k, name in enumerate(["out", "h1", "h2", "c"]):
    r = lambda x: f"{int(((x - t0) * 10).min()):6d} .. {int(((x - t0) * 10).max()):6d}"
    print(f"  {name:4s} flag raised {r(tm[:, k, 0:4])} | all seen {r(tm[:, k, 4:8])} | loads + stores issued {r(tm[:, k, 8:12])} | barrier {r(tm[:, k, 12])} | landed {r(tm[:, k, 13])} | in LDS {r(tm[:, k, 14])}")
    w = [int(((tm[:, k, 8 + i] - tm[:, k, 4 + i]) * 10).float().mean()) for i in range(4)]
    print(f"       seen -> issued, mean per wave: {w}")
sb = m.get_tensor("dc_bstamps").view(torch.int64).cpu().tolist()
bn = ["dpre + raise/wait", "fetch dpre", "dcat product", "dc raise/wait", "attention bwd", "dq raise/wait", "fetch dq", "dh2 product + cell 2", "dz2 raise/wait", "fetch dz2", "dz2 products + cell 1", "dz1 raise/wait", "fetch dz1", "dz1 products"]
tb = sum(sb[:14])
print("backward kernel, workgroup 0:")
for n, v in zip(bn, sb):
    print(f"  {n:24s} {v / L:9.0f} cycles/step  {100.0 * v / max(tb, 1):5.1f} %")
print("  total cycles/step", tb / L)
