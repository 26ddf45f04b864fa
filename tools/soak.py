"""Soak test of the whole-sequence (cluster) kernels: many train steps and decode calls back to back, then aocr_cluster_status.
A group member that ever gave up waiting (bounded spins) would show up as a non-zero code."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "torch-attention-ocr_amd"))
import torch
from test_step_gpu import make
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0            # round 3: the dropout instances of the decoder cluster kernels
beam = int(sys.argv[3]) if len(sys.argv) > 3 else 1                # round 5: > 1 adds a beam-search decode (chain kernel, several launches) every 50th step
for (B, W, L, He, Le) in ((256, 256, 24, 256, 1), (70, 416, 13, 256, 1), (16, 1024, 13, 512, 2)):       # the third: a stacked encoder as a layer wavefront on its own streams (T = 255: 5 chunks)
    m, O, ocfg, P, st, batch = make(dict(enc_hidden=He, enc_layers=Le, dec_layers=2, input_feed=True), B=B, W=W, maxlen=L - 1, compute="bf16", max_decoder_l=50, max_beam=max(beam, 1))
    images, targets, targets_eval = m._upload(batch)
    m.optim_state["learningRate"] = 1e-4
    m.dropout = drop
    t0 = time.time()
    for i in range(steps):
        m.train_step_device(images, targets, targets_eval)
        if i % 10 == 0:
            m.decode_device(images, targets, targets_eval, 1)
        if beam > 1 and i % 50 == 0:
            m.decode_device(images, targets, targets_eval, beam)
    torch.cuda.synchronize(); m.check_health()
    print(f"B={B} W={W}: {steps} train steps + {steps // 10} decode calls in {time.time() - t0:.1f} s, dropout {drop}, beam {beam}, cluster status 0, loss {float(m._scal[0].item()):.3f}")
    m.shutdown()
