"""Time a beam-k decode call at the C3 shape: python tools/debug/beam_time.py [beam] [B] [W]  (AOCR_NO_DEC_CHAINS_BEAM=1 -> per-step launch chain)"""
import sys, time, torch
sys.path.insert(0, "torch-attention-ocr_amd")
import aocr, aocr.synth
k = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
m = aocr.Model().create(dict(encoder_num_hidden=256, encoder_num_layers=1, decoder_num_layers=2, input_feed=True, batch_size=B, max_img_w=W,
                             max_decoder_l=50, max_beam=k, compute="bf16", learning_rate=0.1, seed=910820))
img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=1234, max_len=49, H=32)
images = torch.from_numpy(img).to(device=dev, dtype=torch.float32)
targets = torch.from_numpy(tgt).to(dev); targets_eval = torch.from_numpy(tge).to(dev)
m.decode_device(images, targets, targets_eval, k); torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    m.decode_device(images, targets, targets_eval, k)
torch.cuda.synchronize()
e = (time.perf_counter() - t0) / n
print(f"beam {k} B={B} W={W}: {1e3 * e:.3f} ms/call, {B * 50 / e / 1e6:.2f} M chars/s, cl_err {int(m.get_tensor('cl_err').view(torch.int32)[0])}")
