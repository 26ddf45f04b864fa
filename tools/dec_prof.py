import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "torch-attention-ocr_amd"))
import torch, aocr
m = aocr.Model().create(dict(encoder_num_hidden=256, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=256,
                             max_img_w=256, max_decoder_l=50, max_beam=1, compute="bf16", learning_rate=0.1, seed=910820))
img, tgt, tge, nnz = aocr.synth.synth_batch(256, 256, seed=1234, max_len=23)
dev = m.device
images = torch.from_numpy(img).to(device=dev, dtype=torch.float32); targets = torch.from_numpy(tgt).to(dev); te = torch.from_numpy(tge).to(dev)
for _ in range(3): m.decode_device(images, targets, te, 1)
torch.cuda.synchronize()
m.shutdown()
