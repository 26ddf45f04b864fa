"""BASELINE.json configs[4] at full geometry (128x1024 strips, 2-layer BiLSTM(512), beam 5) -- runs one train step and one beam
decode on the GPU and prints timings; no oracle at this size (parity of the same structure: tests/test_step_gpu.py::test_tall_strips...)."""
import sys, time
sys.path.insert(0, "torch-attention-ocr_amd")
import numpy as np, torch
import aocr
B, H, W, L = 16, 128, 1024, 24
m = aocr.Model().create(dict(encoder_num_hidden=512, encoder_num_layers=2, decoder_num_layers=2, input_feed=True, batch_size=B, img_h=H,
                             max_img_w=W, max_decoder_l=50, max_beam=5, compute="bf16", learning_rate=0.1, seed=1))
img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=5, max_len=L - 1, H=H)
batch = [img, tgt, tge, nnz, [str(i) for i in range(B)]]
for phase in ("train", "decode"):
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if phase == "train":
            loss, _ = m.step(batch, False)
        else:
            loss, stats = m.step(batch, True, 5)
        torch.cuda.synchronize()
        print(f"{phase} {it}: {1e3 * (time.perf_counter() - t0):.1f} ms, loss/token {loss / nnz:.4f}", flush=True)
print("workspace GB", m.workspace.numel() / 2**30)
