#!/usr/bin/env python
"""bench.py -- headline benchmark of the CNN -> BiLSTM -> attention-decoder hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run, one rank per GPU).
A "step" = one full train step of the reference's feval + optim.sgd_list (forward, hand-ordered BPTT, clip, SGD
[+ RCCL gradient all-reduce]) on one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs): c3 (default) = 32x256 crops, batch 256 per GPU, He=256, Ld=2, input feed, L=24,
bf16 operands / fp32 accumulate; c2 = 32x100, batch 64, fp32 MFMA.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "torch-attention-ocr_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    "c3": dict(B=256, W=256, L=24, He=256, Le=1, Ld=2, compute="bf16", name="32x256 crops, batch 256/GPU, VGG-7 + BiLSTM(256) + 2-layer attn decoder, L=24"),
    "c2": dict(B=64, W=100, L=24, He=256, Le=1, Ld=2, compute="f32", name="32x100 crops, batch 64/GPU, VGG-7 + BiLSTM(256) + 2-layer attn decoder, L=24"),
}


def flops_per_image(W, He, Le, Ld, L, E=20, V=39):
    """2*M*N*K per contraction, forward; train = 3x."""
    Hd = 2 * He
    T = W // 4 - 1
    cnn = 2 * (9 * 64 * 32 * W + 9 * 64 * 128 * 8 * W + 9 * 128 * 256 * 2 * W + 9 * 256 * 256 * 2 * W + 9 * 256 * 512 * W
               + 9 * 512 * 512 * W + 4 * 512 * 512 * T)
    enc = 2 * T * sum(2 * ((512 if l == 0 else He) + He) * 4 * He for l in range(Le))
    dec = L * (sum(2 * ((E + Hd if l == 0 else Hd) + Hd) * 4 * Hd for l in range(Ld)) + 6 * Hd * Hd + 4 * T * Hd + 2 * Hd * V)
    return cnn + enc + dec


def _cpu_baseline_worker(wl, threads, seconds_budget):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_torch as O
    torch.set_num_threads(threads)
    cfg = O.OcrConfig(enc_hidden=wl["He"], enc_layers=wl["Le"], dec_layers=wl["Ld"], input_feed=True)
    P = {k: v.float() for k, v in O.init_params(cfg, 910820).items()}
    st = {k: v.float() for k, v in O.init_bn_state().items()}
    Bc = 8
    img, tgt, tge, _ = O.synth_batch(Bc, wl["W"], max_len=wl["L"] - 1)
    img = torch.from_numpy(img).float(); tgt = torch.from_numpy(tgt); tge = torch.from_numpy(tge)
    loss, G, _, _ = O.train_step_manual(P, st, cfg, img, tgt, tge)          # untimed warm-up step (thread pool, allocator)
    t0 = time.time(); n = 0
    while True:
        loss, G, _, _ = O.train_step_manual(P, st, cfg, img, tgt, tge)
        O.sgd_list(P, G, 0.1)
        n += 1
        el = time.time() - t0
        if el > seconds_budget or n >= 64:               # about 10 s of wall time on 16 threads
            break
    print(json.dumps({"value": Bc * n / el, "unit": "image-lines/s", "cores": threads, "kind": "port",
                      "sample": f"{n} train steps of batch {Bc} at 32x{wl['W']} (torch-CPU fp32 restatement of the reference op order, "
                                f"{threads} threads of {os.cpu_count()} host cores)"}))


def cpu_baseline(wl, seconds_budget=10.0):
    """The oracle (torch CPU restatement, fp32) timed on a bounded sample of the same workload, in a child process
    with a hard timeout so the default run always ends within minutes.  16 threads: the per-timestep LSTM ops are
    tiny and slow down badly when a 256-core host is oversubscribed."""
    import subprocess
    threads = min(16, os.cpu_count() or 1)
    code = ("import json,sys; sys.argv=['bench.py']; import importlib.util as u; s=u.spec_from_file_location('bench', %r); "
            "b=u.module_from_spec(s); s.loader.exec_module(b); b._cpu_baseline_worker(json.loads(%r), %d, %f)"
            % (os.path.join(ROOT, "bench.py"), json.dumps(wl), threads, seconds_budget))
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:                                   # timeout or failure: report it, never stall the benchmark
        return {"value": None, "unit": "image-lines/s", "cores": threads, "kind": "port", "sample": f"not measured: {type(e).__name__}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--compute", default=None, choices=["f32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-steps", type=int, default=3)
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 fp32 secondary measurement")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.compute:
        wl["compute"] = args.compute

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AOCR_BENCH_ONE_GPU"):            # debugging aid: all ranks on device 0 (with AOCR_BENCH_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    dist = torch.distributed
    if world > 1:
        backend = os.environ.get("AOCR_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    import aocr
    B, W, L = wl["B"], wl["W"], wl["L"]
    m = aocr.Model().create(dict(encoder_num_hidden=wl["He"], encoder_num_layers=wl["Le"], decoder_num_layers=wl["Ld"],
                                 input_feed=True, batch_size=B, max_img_w=W, max_decoder_l=50, max_beam=1,
                                 compute=wl["compute"], learning_rate=0.1, seed=910820))
    if world > 1:
        dist.broadcast(m.params, 0)                     # identical replicas
    img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=1234 + rank, max_len=L - 1)
    dev = m.device
    images = torch.from_numpy(img).to(device=dev, dtype=torch.float32)
    targets = torch.from_numpy(tgt).to(dev); targets_eval = torch.from_numpy(tge).to(dev)
    assert targets.shape[1] == L

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        m.train_step_device(images, targets, targets_eval)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = m.train_step_device(images, targets, targets_eval)
    sync()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    loss_val = float(loss.item())
    lines_per_s = world * B * args.steps / el
    replica_drift = None
    if world > 1:                                       # every rank must hold the same parameters after the timed steps
        cs = m.params.double().abs().sum().reshape(1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        replica_drift = float((hi - lo).item())

    # decode throughput (greedy, max_decoder_l = 50 steps + gold pass, the reference's -phase test step)
    dec = None; dec_dict = None
    if args.decode_steps > 0:
        m.decode_device(images, targets, targets_eval, 1); sync()
        t0 = time.perf_counter()
        for _ in range(args.decode_steps):
            m.decode_device(images, targets, targets_eval, 1)
        sync()
        eld = time.perf_counter() - t0
        dec = world * B * 50 * args.decode_steps / eld
        # the same step under -use_dictionary: a synthetic 90 k-word lexicon as a device-resident flat trie,
        # admissibility tested inside the selection kernel (the reference walks Lua tables per image, beam and candidate)
        rng = np.random.default_rng(1234)
        lens = rng.integers(3, 11, size=90000)
        letters = rng.integers(0, 26, size=int(lens.sum())).astype(np.uint8) + 97
        words, o = [], 0
        for n in lens:
            words.append(letters[o:o + n].tobytes().decode()); o += int(n)
        trie = aocr.build_trie(words).to(dev)
        m.decode_device(images, targets, targets_eval, 1, trie); sync()
        t0 = time.perf_counter()
        for _ in range(args.decode_steps):
            m.decode_device(images, targets, targets_eval, 1, trie)
        sync()
        dec_dict = {"chars_per_s": world * B * 50 * args.decode_steps / (time.perf_counter() - t0), "words": len(words),
                    "trie_nodes": trie.n_nodes, "trie_bytes": int(trie.mask.nbytes + trie.base.nbytes + trie.child.nbytes)}

    # secondary line (N = 1 only): BASELINE.json configs[1] = C2 in exact-fp32 MFMA mode (the 1e-4 logit-parity configuration)
    c2 = None
    if world == 1 and args.workload == "c3" and not args.no_secondary:
        w2 = WORKLOADS["c2"]
        m2 = aocr.Model().create(dict(encoder_num_hidden=w2["He"], encoder_num_layers=w2["Le"], decoder_num_layers=w2["Ld"],
                                      input_feed=True, batch_size=w2["B"], max_img_w=w2["W"], max_decoder_l=50, max_beam=1,
                                      compute="f32", learning_rate=0.1, seed=910820))
        i2, t2, e2, _ = aocr.synth.synth_batch(w2["B"], w2["W"], seed=1234, max_len=w2["L"] - 1)
        i2 = torch.from_numpy(i2).to(device=dev, dtype=torch.float32); t2 = torch.from_numpy(t2).to(dev); e2 = torch.from_numpy(e2).to(dev)
        for _ in range(3):
            m2.train_step_device(i2, t2, e2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            m2.train_step_device(i2, t2, e2)
        torch.cuda.synchronize(); e = time.perf_counter() - t0
        c2 = {"workload": "c2: " + w2["name"], "dtype": "f32", "steps": 10, "ms_per_step": 1e3 * e / 10, "value": w2["B"] * 10 / e,
              "unit": "image-lines/s"}
        m2.shutdown()

    # data path: 255*rgb2y + image.scale of one C3 batch of decoded 48x384 RGB line images to 32x256,
    # inputs resident in HBM; HBM-bound, reported against the 8 TB/s peak (algorithmic bytes = uint8 source + fp32 output)
    dp = None
    if world == 1 and not args.no_secondary:
        import ctypes as C
        from aocr.data import ImageDesc
        n, sh, sw, ow = B, 48, 384, W
        src = torch.randint(0, 256, (n * sh * sw * 3,), dtype=torch.uint8, device=dev)
        desc = (ImageDesc * n)(*[ImageDesc(i * sh * sw * 3, sh, sw, 3, 0) for i in range(n)])
        dsc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
        outp = torch.empty((n, 1, 32, ow), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        run = lambda: aocr.check(aocr.lib.aocr_preprocess_lines(st, aocr.ptr(src), aocr.ptr(dsc), n, 32, ow, aocr.ptr(outp)))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        msd = e0.elapsed_time(e1) / 20
        byts = n * (sh * sw * 3 + 32 * ow * 4)
        dp = {"kernel": "aocr_preprocess_lines (rgb2y + image.scale 48x384x3 -> 32x%d)" % ow, "images_per_s": n / (msd * 1e-3),
              "ms_per_batch": msd, "bound": "hbm", "achieved_GBps": byts / (msd * 1e-3) / 1e9, "peak_GBps": 8000.0,
              "frac": byts / (msd * 1e-3) / 1e9 / 8000.0}

    out = None
    if rank == 0:
        bf16 = wl["compute"] == "bf16"
        ms, fl = m.profile_kernel(0, 20)
        peak = 2500.0 if bf16 else 157.3
        ach = fl / (ms * 1e-3) / 1e12
        fpi = flops_per_image(W, wl["He"], wl["Le"], wl["Ld"], L)
        traffic = None                                   # PMC passes cannot run inside this process: value measured with
        pmc = os.path.join(ROOT, "profiles", "r01_conv6_fwd_pmc.json")       # rocprofv3 --pmc (tools/pmc_traffic.py), C3 bf16 only
        if bf16 and args.workload == "c3" and os.path.exists(pmc):
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
        out = {
            "metric": "image-lines/sec (train step)", "value": lines_per_s, "unit": "image-lines/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['name']}", "global_batch": world * B, "img": f"32x{W}",
                       "decoder_steps": L, "parallelism": f"dp{world}", "input_feed": True},
            "train_gflop_per_image": 3 * fpi / 1e9, "step_tflops": 3 * fpi * lines_per_s / 1e12,
            "step_mfma_frac": 3 * fpi * lines_per_s / 1e12 / (peak * world),
            "decode_chars_per_s": dec, "decode_dict": dec_dict, "replica_drift": replica_drift, "loss": loss_val, "secondary": c2, "data_path": dp,
            "roofline": {"bound": "mfma", "kernel": "conv6 forward implicit GEMM (512->512 3x3 + ReLU + pool)",
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                         "ms_per_launch": ms},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl)
    m.shutdown()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
