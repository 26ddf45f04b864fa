"""AOCR_CL_STAMP=1 python tools/cl_stamp.py : cycle stamps of one wave of the cluster encoder forward kernel at C3 shape."""
import os, sys
os.environ["AOCR_CL_STAMP"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-attention-ocr_amd"))
import torch, aocr
B, W, L = 256, 256, 24
m = aocr.Model().create(dict(encoder_num_hidden=256, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=B, max_img_w=W,
                             max_decoder_l=50, max_beam=1, compute="bf16", learning_rate=0.1, seed=1))
img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=5, max_len=L - 1)
images = torch.from_numpy(img).cuda().float(); targets = torch.from_numpy(tgt).cuda(); te = torch.from_numpy(tge).cuda()
for _ in range(4):
    m.train_step_device(images, targets, te)
torch.cuda.synchronize()
