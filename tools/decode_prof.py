"""Decode-only loop at the C3 shape for `rocprofv3 --kernel-trace --stats` (tools/decode_stats.sh): which launches make up a -phase test call."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-attention-ocr_amd"))
import aocr

B, W, L = 256, 256, 24
m = aocr.Model().create(dict(encoder_num_hidden=256, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=B, max_img_w=W,
                             max_decoder_l=50, max_beam=1, compute="bf16", learning_rate=0.1, seed=910820))
img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=1234, max_len=L - 1)
dev = m.device
images = torch.from_numpy(img).to(device=dev, dtype=torch.float32); targets = torch.from_numpy(tgt).to(dev); targets_eval = torch.from_numpy(tge).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    m.decode_device(images, targets, targets_eval, 1)
torch.cuda.synchronize()
m.shutdown()
