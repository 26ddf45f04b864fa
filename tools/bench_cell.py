"""Micro-benchmark of the fused LSTM-cell (gates) kernel through the C ABI: time vs K to separate fixed cost from the K loop."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "torch-attention-ocr_amd"))
import torch, aocr
def run(B, inp, H, compute, iters=200):
    d = "cuda"
    x = torch.randn(B, inp, device=d); h = torch.randn(B, H, device=d); c = torch.randn(B, H, device=d)
    Wi = torch.randn(4 * H, inp, device=d) * 0.05; bi = torch.zeros(4 * H, device=d); Wh = torch.randn(4 * H, H, device=d) * 0.05; bh = torch.zeros(4 * H, device=d)
    co = torch.empty(B, H, device=d); ho = torch.empty(B, H, device=d); g = torch.empty(B, 4 * H, device=d)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def go():
        aocr.check(aocr.lib.aocr_lstm_cell_forward(st, compute, aocr.ptr(x), inp, aocr.ptr(h), aocr.ptr(c), aocr.ptr(Wi), aocr.ptr(bi), aocr.ptr(Wh), aocr.ptr(bh), aocr.ptr(co), aocr.ptr(ho), aocr.ptr(g), B, H))
    for _ in range(10): go()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B, inp, H in [(256, 16, 256), (256, 256, 256), (256, 768, 256), (256, 1792, 256), (256, 512, 512), (64, 512, 512), (32, 16, 32)]:
    print(f"B={B} in={inp} H={H} K={inp+H}: f32 {run(B, inp, H, 0):7.1f} us   bf16 {run(B, inp, H, 1):7.1f} us")
