import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "torch-attention-ocr_amd"); sys.path.insert(0, "oracle")
import torch
from test_step_gpu import make
from aocr import check, lib, ptr
from aocr import dist as adist
m, O, ocfg, P, st, batch = make(dict(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True), B=32, W=100, maxlen=10, max_decoder_l=12, max_beam=1, compute="bf16")
images, targets, targets_eval = m._upload(batch)
B, _, _, W = images.shape
loss = torch.zeros(1, device="cuda")
def enqueue():
    check(lib.aocr_train_forward_backward(m._h, ptr(images), ptr(targets), ptr(targets_eval), B, W, targets.shape[1], 1.0 / B, ptr(loss)))
check(lib.aocr_model_set_stream(m._h, m._stream()))
enqueue(); torch.cuda.synchronize()
g0 = m.grad_params.clone()
ranges = adist.bucket_ranges(m.ccfg)
adist.world_size = lambda: 2
torch.distributed.all_reduce = lambda t: t.mul_(2.0)
comm = torch.cuda.Stream()
for name, wb in (("with waits", m._wait_bucket), ("NO waits", lambda k, s: None)):
    enqueue()
    adist.exchange_overlapped(m.grad_params, loss, ranges, wb, comm)
    g = m.grad_params.clone(); torch.cuda.synchronize()
    print(name, "max |g - 2 g0| / max|g0| =", ((g - 2 * g0).abs().max() / g0.abs().max()).item())
