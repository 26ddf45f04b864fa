"""Run only aocr_profile_kernel (conv6 forward at the C3 shape) -- the target of rocprofv3 --pmc passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "torch-attention-ocr_amd"))
import torch
import aocr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = aocr.Model().create(dict(encoder_num_hidden=256, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=256,
                             max_img_w=256, max_decoder_l=50, max_beam=1, compute="bf16", learning_rate=0.1, seed=910820))
img, tgt, tge, nnz = aocr.synth.synth_batch(256, 256, seed=1234, max_len=23)
dev = m.device
m.train_step_device(torch.from_numpy(img).to(device=dev, dtype=torch.float32), torch.from_numpy(tgt).to(dev), torch.from_numpy(tge).to(dev))
torch.cuda.synchronize()
ms, fl = m.profile_kernel(0, reps)
print(f"conv6 fwd {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s")
m.shutdown()
