"""Where the C++ restatement (oracle/cpu_ref) spends a train step on this host: AOCR_CPU_REF_TIMING=1 python tools/cpu_ref_timing.py [B] [W] [threads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3: os.environ["OMP_NUM_THREADS"] = sys.argv[3]
os.environ.setdefault("AOCR_CPU_REF_TIMING", "1")
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "oracle", "cpu_ref"))
import numpy as np, oracle_torch as O, cpu_ref as R
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32; W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = O.OcrConfig(enc_hidden=256, enc_layers=1, dec_layers=2, input_feed=True)
names = [s[0] for s in O.param_spec(cfg)]
flat = R.flatten({k: v.numpy() for k, v in O.init_params(cfg).items()}, names)
bn = np.concatenate([np.zeros(256), np.ones(256), np.zeros(512), np.ones(512), np.zeros(512), np.ones(512)])
img, tgt, tge, _ = O.synth_batch(B, W, max_len=23)
rc = R.make_cfg(256, 1, 2, True)
for dt in (np.float32, np.float32, np.float64):
    t0 = time.time(); R.train_step(rc, flat, bn, img, tgt, tge, dtype=dt); el = time.time() - t0
    print(f"{dt.__name__} B={B} W={W} threads={R.threads()}: {el:.2f} s = {B / el:.2f} lines/s", flush=True)
