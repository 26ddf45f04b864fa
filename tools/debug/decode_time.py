"""Time decode calls of any shape: python tools/debug/decode_time.py beam B W H He Le [calls]   (e.g. 5 256 1024 128 512 2 = BASELINE config 5; 1 400 100 32 512 1 = the reference default)
Under `rocprofv3 --kernel-trace --stats -- python3 tools/debug/decode_time.py ...` the per-kernel table is the decode call's alone."""
import sys, time, torch
sys.path.insert(0, "torch-attention-ocr_amd")
import aocr, aocr.synth
k, B, W, H, He, Le = (int(x) for x in sys.argv[1:7])
n = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0")
m = aocr.Model().create(dict(encoder_num_hidden=He, encoder_num_layers=Le, decoder_num_layers=2, input_feed=True, batch_size=B, max_img_w=W, img_h=H,
                             max_decoder_l=50, max_beam=max(k, 1), compute="bf16", learning_rate=0.1, seed=910820))
img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=1234, max_len=23, H=H)
images = torch.from_numpy(img).to(device=dev, dtype=torch.float32)
targets = torch.from_numpy(tgt).to(dev); targets_eval = torch.from_numpy(tge).to(dev)
m.decode_device(images, targets, targets_eval, k); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    m.decode_device(images, targets, targets_eval, k)
torch.cuda.synchronize()
e = (time.perf_counter() - t0) / n
print(f"beam {k} B={B} {H}x{W} He={He} Le={Le}: {1e3 * e:.3f} ms/call, {B * 50 / e / 1e6:.3f} M chars/s (nominal 50 steps)")
