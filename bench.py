#!/usr/bin/env python
"""bench.py -- headline benchmark of the CNN -> BiLSTM -> attention-decoder hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W`  (N>1: launched by torch.distributed.run, one rank per GPU; a bare
`python bench.py --gpus N` starts that launcher itself as a child process, before anything touches a GPU).
A "step" = one full train step of the reference's feval + optim.sgd_list (forward, hand-ordered BPTT, clip, SGD
[+ RCCL gradient all-reduce]) on one synthetic batch already resident in HBM.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs): c3 (default) = 32x256 crops, batch 256 per GPU, He=256, Ld=2, input feed, L=24,
bf16 operands / fp32 accumulate; c2 = 32x100, batch 64, fp32 MFMA; ref = the reference's own default shape
(src/train.lua:41,47: batch 400, He=512) in bf16.
--scaling weak (default): the per-GPU batch is fixed; strong: the GLOBAL batch is the workload's batch, split over the ranks.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "torch-attention-ocr_amd"))

WORKLOADS = {
    "c3": dict(B=256, W=256, L=24, He=256, Le=1, Ld=2, compute="bf16", name="32x256 crops, batch 256/GPU, VGG-7 + BiLSTM(256) + 2-layer attn decoder, L=24"),
    # the strong-scaling slice of C3: BASELINE's "batch 256" read as the GLOBAL batch of an 8-GPU job = 32 lines per GPU
    "c3s": dict(B=32, W=256, L=24, He=256, Le=1, Ld=2, compute="bf16", name="32x256 crops, batch 32/GPU (C3's global batch 256 over 8 GPUs), VGG-7 + BiLSTM(256) + 2-layer attn decoder, L=24"),
    "c2": dict(B=64, W=100, L=24, He=256, Le=1, Ld=2, compute="f32", name="32x100 crops, batch 64/GPU, VGG-7 + BiLSTM(256) + 2-layer attn decoder, L=24"),
    "ref": dict(B=400, W=100, L=24, He=512, Le=1, Ld=2, compute="bf16", name="32x100 crops, batch 400/GPU, VGG-7 + BiLSTM(512) + 2-layer attn decoder (train.lua defaults), L=24"),
    # BASELINE.json configs[3]: variable-width crops 32x{64..800} in width buckets (data_gen.lua:91-99 emits one width per batch), 64 lines per GPU
    # (global batch 512 on 8 GPUs); every step draws its bucket from a fixed seeded sequence, so the FLOPs per step vary: mean reported
    "c4": dict(B=64, W=800, L=24, He=256, Le=1, Ld=2, compute="bf16", widths=list(range(64, 801, 32)),
               name="32x{64..800} crops in 24 width buckets, batch 64/GPU, VGG-7 + BiLSTM(256) + 2-layer attn decoder, L=24"),
    # BASELINE.json configs[4]: 128x1024 full-line strips (the CNN leaves 7 x 255 feature positions: T = 1785), 2-layer BiLSTM(512), beam-5 decode
    # BASELINE names no batch.  Round 6: 256 strips per GPU -- the batch at which the encoder's 16-row groups cover the chip (2 directions x 16 groups x 8 CUs; at the
    # 16 strips of rounds 1-5 they kept 16-32 of 256 CUs busy for 88 % of the step: 0.59 k lines/s was a property of the batch).  `--batch 16|64|128` for the others.
    "c5": dict(B=256, W=1024, H=128, L=24, He=512, Le=2, Ld=2, compute="bf16", beam=5, name="128x1024 strips, batch 256/GPU, VGG-7 + 2-layer BiLSTM(512) + 2-layer attn decoder, L=24, beam-5 decode"),
}
PEAK = {"bf16": 2500.0, "f32": 157.3}          # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


def flops(W, He, Le, Ld, L, E=20, V=39, H=32):
    """Algorithmic forward FLOPs per image-line by family (2*M*N*K per contraction).  H > 32 (strips): every map has
    H / 32 times the rows, the 2x2 convolution leaves (H / 16 - 1) x (W / 4 - 1) feature positions."""
    Hd, T = 2 * He, (H // 16 - 1) * (W // 4 - 1)
    hs = H // 32
    conv1 = 2 * 9 * 64 * 32 * W * hs
    convs = 2 * (hs * (9 * 64 * 128 * 8 * W + 9 * 128 * 256 * 2 * W + 9 * 256 * 256 * 2 * W + 9 * 256 * 512 * W + 9 * 512 * 512 * W) + 4 * 512 * 512 * T)
    enc_in = 2 * T * sum(2 * (512 if l == 0 else He) * 4 * He for l in range(Le))        # hoisted input projections
    enc_rec = 2 * T * Le * 2 * He * 4 * He                                                # recurrent h . W_h2h
    dec_emb = L * 2 * E * 4 * Hd                                                          # hoisted embedding part of layer 1
    dec_chain = L * (sum(2 * ((Hd if l == 0 else Hd) + Hd) * 4 * Hd for l in range(Ld)) + 6 * Hd * Hd + 4 * T * Hd)
    proj = L * 2 * Hd * V
    return dict(conv1=conv1, convs=convs, enc_in=enc_in, enc_rec=enc_rec, dec_emb=dec_emb, dec_chain=dec_chain, proj=proj,
                total=conv1 + convs + enc_in + enc_rec + dec_emb + dec_chain + proj)


# ------------------------------------------------------------------------------------------------ CPU baseline (child process)
def _cpu_baseline_worker(wl, budget_s):
    """oracle/cpu_ref: the C++/OpenMP restatement of the reference's CPU op order, fp64 (the reference's CPU tensors)
    and fp32, on ALL host cores, at the workload's full batch when the host gets through it inside the budget."""
    sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "oracle", "cpu_ref"))
    import numpy as np
    import cpu_ref as R
    import oracle_torch as O
    cores = R.threads()
    cfg = O.OcrConfig(enc_hidden=wl["He"], enc_layers=wl["Le"], dec_layers=wl["Ld"], input_feed=True)
    names = [s[0] for s in O.param_spec(cfg)]
    flat = R.flatten({k: v.numpy() for k, v in O.init_params(cfg, 910820).items()}, names)
    bn = np.concatenate([np.zeros(256), np.ones(256), np.zeros(512), np.ones(512), np.zeros(512), np.ones(512)])
    rc = R.make_cfg(wl["He"], wl["Le"], wl["Ld"], True)

    def run(B, dtype):
        img, tgt, tge, _ = O.synth_batch(B, wl["W"], max_len=wl["L"] - 1)
        t0 = time.time()
        r = R.train_step(rc, flat, bn, img, tgt, tge, dtype=dtype)
        R.sgd(rc, flat, r["grads"], 0.1, 5.0, dtype=dtype)
        return time.time() - t0
    Bc = min(wl["B"], max(8, cores // 4))
    run(Bc, np.float32)                                      # warm-up (thread pool, page faults)
    t_cal = run(Bc, np.float32)                              # calibration: lines/s at a small batch
    est_full = t_cal / Bc * wl["B"] * 1.6                    # fp32 + fp64 pass at the full batch, fp64 ~ 1.6 x
    B = wl["B"] if est_full * 2 < budget_s else max(Bc, int(wl["B"] * budget_s / (est_full * 2)) // 8 * 8)
    t32 = run(B, np.float32)
    t64 = run(B, np.float64)
    # greedy decode of the same lines (50 steps + gold pass), fp64, bounded batch
    Bd = min(B, 64)
    img, tgt, tge, _ = O.synth_batch(Bd, wl["W"], max_len=wl["L"] - 1)
    t0 = time.time(); R.decode(rc, flat, bn, img, tgt, tge, 1, 50); td = time.time() - t0
    # the same port on BASELINE.json configs[1] (C2: 32x100, batch 64, fp32 and fp64) -- SURVEY.md 8(d) asks for the small shapes beside C3
    c2 = None
    if wl["W"] != 100:
        w2 = dict(wl, W=100)
        def run2(dtype):
            img2, tgt2, tge2, _ = O.synth_batch(64, 100, max_len=wl["L"] - 1)
            t0 = time.time()
            r = R.train_step(rc, flat, bn, img2, tgt2, tge2, dtype=dtype)
            R.sgd(rc, flat, r["grads"], 0.1, 5.0, dtype=dtype)
            return time.time() - t0
        a32, a64 = run2(np.float32), run2(np.float64)
        c2 = {"workload": "c2: 32x100 crops, batch 64", "value_f32": 64 / a32, "value_f64": 64 / a64, "unit": "image-lines/s", "seconds": [a32, a64]}
    print(json.dumps({"value": B / t64, "unit": "image-lines/s", "cores": cores, "kind": "port",
                      "impl": "cpp-restatement: oracle/cpu_ref (im2col+GEMM conv, per-timestep LSTM, unfused attention, OpenMP), checked "
                              "against tests/golden in the CPU suite",
                      "dtype": "f64", "batch": B, "value_f32": B / t32, "decode_chars_per_s_f64": Bd * 50 / td,
                      "sample": f"1 train step (forward + BPTT + clip + SGD) of batch {B} at 32x{wl['W']} in fp64 ({t64:.1f} s) and one in fp32 "
                                f"({t32:.1f} s), after a warm-up step of batch {Bc}; greedy decode of {Bd} lines ({td:.1f} s); "
                                f"{cores} OpenMP threads on the {len(os.sched_getaffinity(0))} cores this process may run on (os.cpu_count() = {os.cpu_count()})",
                      "c2_shape": c2}))


def cpu_baseline(wl, budget_s=40.0):
    code = ("import json,sys; sys.argv=['bench.py']; import importlib.util as u; s=u.spec_from_file_location('bench', %r); "
            "b=u.module_from_spec(s); s.loader.exec_module(b); b._cpu_baseline_worker(json.loads(%r), %f)"
            % (os.path.join(ROOT, "bench.py"), json.dumps(wl), budget_s))
    n = len(os.sched_getaffinity(0)) or 1                   # the cores this process may actually run on, not the host's total
    env = dict(os.environ, OMP_NUM_THREADS=str(n), OMP_PROC_BIND="false", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:                                   # timeout or failure: report it, never stall the benchmark
        return {"value": None, "unit": "image-lines/s", "cores": n, "kind": "port", "sample": f"not measured: {type(e).__name__}"}


# ------------------------------------------------------------------------------------------------ main
def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--compute", default=None, choices=["f32", "bf16"])
    ap.add_argument("--batch", type=int, default=None, help="lines per GPU instead of the workload's own (c5: BASELINE names no batch; 16 / 64 / 128 are reported)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-steps", type=int, default=20)
    ap.add_argument("--no-secondary", action="store_true", help="skip the C2 fp32 and data-path secondary measurements")
    ap.add_argument("--dropout", type=float, default=0.0, help="-dropout of the reference (LSTM.lua:68-69,116-118); 0 = the reference default")
    ap.add_argument("--provider", default="callback", choices=["callback", "rccl"],
                    help="N > 1: who sums the gradients -- torch.distributed through aocr_comm_set_callback, or the library's own RCCL binding (aocr_comm_init_rank)")
    ap.add_argument("--sustain-seconds", type=float, default=8.0, help="keep stepping after the timed steps until this much GPU time has passed")
    return ap.parse_args()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # not under a launcher: start one as a CHILD (never exec from a process that may touch the GPU) and relay its output
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-GPU run as {args.gpus} GPUs")

    import numpy as np
    import torch
    wl = dict(WORKLOADS[args.workload])
    if args.compute:
        wl["compute"] = args.compute
    if args.batch:
        wl["name"] = wl["name"].replace(f"batch {wl['B']}/GPU", f"batch {args.batch}/GPU"); wl["B"] = args.batch
    rank = int(os.environ.get("RANK", "0")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AOCR_BENCH_WATCHDOG"):          # debugging aid: dump every thread's stack and exit if the run exceeds N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["AOCR_BENCH_WATCHDOG"]), exit=True)
    if os.environ.get("AOCR_BENCH_ONE_GPU"):            # debugging aid: all ranks on device 0 (with AOCR_BENCH_BACKEND=gloo)
        local = 0
        # The whole-sequence ("cluster") kernels need the device to themselves: one launch occupies every compute unit and its groups
        # wait for each other, so two ranks' launches interleaved on ONE device starve each other until the bounded waits give up
        # (aocr_cluster_status).  Ranks that share a device run the per-step launch chains instead.
        os.environ.setdefault("AOCR_NO_CLUSTER", "1"); os.environ.setdefault("AOCR_NO_DEC_CLUSTER", "1")
    torch.cuda.set_device(local)
    dist = torch.distributed
    if world > 1:
        backend = os.environ.get("AOCR_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    import aocr
    global_B = wl["B"] * world if args.scaling == "weak" else wl["B"]
    assert global_B % world == 0, "strong scaling needs the global batch to divide by the rank count"
    B, W, L = global_B // world, wl["W"], wl["L"]
    IMG_H, BEAM = wl.get("H", 32), wl.get("beam", 1)
    m = aocr.Model().create(dict(encoder_num_hidden=wl["He"], encoder_num_layers=wl["Le"], decoder_num_layers=wl["Ld"],
                                 input_feed=True, batch_size=B, img_h=IMG_H, max_img_w=W, max_decoder_l=50, max_beam=BEAM,
                                 compute=wl["compute"], learning_rate=0.1, seed=910820))
    if args.dropout > 0:
        m.dropout = float(args.dropout)                                  # masks advance with every step (Model._arm_dropout)
    rccl_ranks = None
    if world > 1:
        if args.provider == "rccl":                                      # the library's own binding: ncclCommInitRank + ncclCommSplit inside libaocr
            aocr.dist.attach_rccl(m, sync_bn=not os.environ.get("AOCR_NO_SYNC_BN"))
        dist.broadcast(m.params, 0); dist.broadcast(m.bn_state, 0)       # identical replicas (parameters and running statistics)
        probe = torch.ones(1, device=m.device); dist.all_reduce(probe)   # an actual collective: how many ranks does it span?
        rccl_ranks = int(probe.item())
    img, tgt, tge, nnz = aocr.synth.synth_batch(B, W, seed=1234 + rank, max_len=L - 1, H=IMG_H)
    dev = m.device
    params0, bn0 = m.params.clone(), m.bn_state.clone()      # the replica the decode legs run on (decode must not depend on how long the training loop ran)
    images = torch.from_numpy(img).to(device=dev, dtype=torch.float32)
    targets = torch.from_numpy(tgt).to(dev); targets_eval = torch.from_numpy(tge).to(dev)
    assert targets.shape[1] == L

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bucket_imgs = None
    if wl.get("widths"):                                # c4: one pre-generated batch per width bucket, visited in a fixed seeded order
        order = np.random.default_rng(4321).permutation(len(wl["widths"]))
        bucket_imgs = []
        for wi in order:
            bi, _, _, _ = aocr.synth.synth_batch(B, wl["widths"][wi], seed=1234 + rank, max_len=L - 1, H=IMG_H)
            bucket_imgs.append(torch.from_numpy(bi).to(device=dev, dtype=torch.float32))
    step_no = [0]

    def step():
        if bucket_imgs is not None:
            im = bucket_imgs[step_no[0] % len(bucket_imgs)]; step_no[0] += 1
            return m.train_step_device(im, targets, targets_eval)
        return m.train_step_device(images, targets, targets_eval)

    def healthy() -> bool:
        """False if a whole-sequence kernel of ANY rank gave up waiting for its group since the last call (results invalid)."""
        code = torch.tensor([m.cluster_status()], device=dev, dtype=torch.int32)
        if world > 1:
            dist.all_reduce(code, op=dist.ReduceOp.MAX)
        return int(code.item()) == 0

    cluster_fallback = False
    for attempt in range(2):
        for _ in range(args.warmup):
            step()
        sync()
        # ---- the timed region: EXACTLY --steps steps between two barriers + device syncs; per-step HIP events ride along on the same stream
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(args.steps):
            loss = step()
            ev[i + 1].record()
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        per_step = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])
        loss_val = float(loss.item())
        if healthy():
            break
        # Something else kept part of the chip busy and a cluster kernel's bounded wait expired: the timed steps are invalid.  Measure
        # again on the per-step launch chains (read per call by the library) and say so in the JSON line.
        assert attempt == 0, "the launch-chain paths cannot time out"
        os.environ["AOCR_NO_CLUSTER"] = "1"; os.environ["AOCR_NO_DEC_CLUSTER"] = "1"; cluster_fallback = True
        if rank == 0:
            print("[bench] a whole-sequence kernel timed out waiting for its group; re-measuring on the launch chains", file=sys.stderr, flush=True)
    lines_per_s = global_B * args.steps / el
    # ---- steady state: keep the GPU busy long enough for an external sampler, report it separately (never `value`)
    sustained = None
    if args.sustain_seconds > 0:
        n = max(args.steps, int(args.sustain_seconds / max(el / args.steps, 1e-4)) + 1)
        sync(); t0 = time.perf_counter()
        for _ in range(n):
            step()
        sync()
        es = time.perf_counter() - t0
        sustained = {"steps": n, "seconds": es, "ms_per_step": 1e3 * es / n, "image_lines_per_s": global_B * n / es, "healthy": healthy()}
    exposed = None
    if world > 1:                                       # part of the gradient exchange the backward pass did not hide, last step (max over ranks)
        ex = torch.tensor([m.comm_exposed_ms()], device=dev, dtype=torch.float64)
        dist.all_reduce(ex, op=dist.ReduceOp.MAX)
        exposed = float(ex.item())
    replica_drift = None
    if world > 1:                                       # every rank must hold the same parameters after the timed steps
        cs = m.params.double().abs().sum().reshape(1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX); dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        replica_drift = float((hi - lo).item())

    fl = flops(W, wl["He"], wl["Le"], wl["Ld"], L, H=IMG_H)
    if wl.get("widths"):                                # mean over the buckets (every bucket is visited equally often over a whole cycle)
        per = [flops(w_, wl["He"], wl["Le"], wl["Ld"], L, H=IMG_H) for w_ in wl["widths"]]
        fl = {k: sum(p_[k] for p_ in per) / len(per) for k in per[0]}
    peak = PEAK[wl["compute"]]
    # ---- per-family HIP-event timing of one step (library marks, include/aocr.h AOCR_PROF_*).  EVERY rank runs the profiled steps (a step
    # contains the gradient all-reduce and the synchronised BatchNorm sums: a rank that skipped them would leave the others waiting);
    # rank 0 reports
    families = None
    fam_all = m.profile_families(step, repeats=3)
    if rank == 0:
        fam = fam_all
        # algorithmic FLOPs per image of each family: the three conv passes; both recurrences of the encoder (h W_h2h forward, dz W_h2h
        # backward); the hoisted GEMMs = input projections / embedding part / projector (forward, d(input), d(weight)) plus the weight
        # gradients of every matrix the two step chains use; the decoder chains themselves (forward = backward in FLOPs)
        alg = {"conv_fwd": fl["convs"], "conv_dgrad": fl["convs"], "conv_wgrad": fl["convs"], "encoder_seq": 2 * fl["enc_rec"],
               "rnn_gemm": 3 * (fl["enc_in"] + fl["dec_emb"] + fl["proj"]) + fl["enc_rec"] + fl["dec_chain"],
               "decoder_fwd": fl["dec_chain"], "decoder_bwd": fl["dec_chain"]}
        families = {}
        for k, ms in fam.items():
            if ms <= 0:
                continue
            e = {"ms_per_step": ms}
            if k in alg:
                tf = alg[k] * B / (ms * 1e-3) / 1e12
                e.update({"algorithmic_gflop": alg[k] * B / 1e9, "tflops": tf, "frac_of_mfma_peak": tf / peak})
            families[k] = e
        families["_sum_ms"] = float(sum(fam.values()))

    # ---- decode throughput (greedy, max_decoder_l = 50 steps + gold pass = the reference's -phase test step), no exchange across ranks.
    # Run on a FRESH replica of the parameters (the ones the training loop started from): after ~1500 steps on one fixed batch the model
    # emits EOS early and the greedy kernel's early exit (every row of a 32-row group at EOS / PAD) skips half the loop -- the number would
    # depend on --sustain-seconds.  Two timings: the kernel as shipped (early exit on) and AOCR_NO_DEC_EARLY=1 = the literal 50 steps of
    # model.lua:376; both with the steps actually executed.
    dec = None; dec_dict = None

    def executed_steps(lab):
        """Decoder steps the greedy kernel ran for these labels: a 32-row group leaves its loop at the first step at which every row has
        emitted EOS / PAD = 1 + the last position with a live token + the EOS step, at most the label width."""
        live = ((lab != 1) & (lab != 3)).cpu().numpy()
        last = np.where(live.any(axis=1), live.shape[1] - np.argmax(live[:, ::-1], axis=1), 0)
        grp = [int(min(lab.shape[1], last[g:g + 32].max() + 2)) for g in range(0, B, 32)]
        return int(sum(min(32, B - g) * grp[g // 32] for g in range(0, B, 32))), max(grp)

    if args.decode_steps > 0:
        trained = (m.params.clone(), m.bn_state.clone())
        m.params.copy_(params0); m.bn_state.copy_(bn0)
        Hd, T = 2 * wl["He"], (IMG_H // 16 - 1) * (W // 4 - 1)
        cluster = wl["compute"] == "bf16" and Hd == 512 and BEAM == 1 and not os.environ.get("AOCR_NO_DEC_CLUSTER")

        def time_decode(no_early):
            if no_early:
                os.environ["AOCR_NO_DEC_EARLY"] = "1"
            try:
                lab = m.decode_device(images, targets, targets_eval, BEAM)[0]; sync()
                t0 = time.perf_counter()
                for _ in range(args.decode_steps):
                    m.decode_device(images, targets, targets_eval, BEAM)
                sync()
                el_ = (time.perf_counter() - t0) / args.decode_steps
                ex, exg = (B * 50, 50) if (no_early or not cluster) else executed_steps(lab)
                r = {"ms_per_call": 1e3 * el_, "chars_per_s": world * B * 50 / el_, "executed_steps_per_call": ex, "executed_steps_per_s": world * ex / el_,
                     "longest_group_steps": exg}
                if rank == 0:
                    fam = m.profile_families(lambda: m.decode_device(images, targets, targets_eval, BEAM), repeats=2)
                    r.update({"beam_ms": fam["decode_chain"], "gold_pass_ms": fam["decoder_fwd"] + fam["rnn_gemm"],
                              "cnn_encoder_ms": sum(fam[k] for k in ("conv_fwd", "bn", "pool_conv1", "encoder_seq", "other")),
                              "beam_us_per_executed_step": 1e3 * fam["decode_chain"] / max(1, exg)})
                return r
            finally:
                os.environ.pop("AOCR_NO_DEC_EARLY", None)

        d_early, d_full = time_decode(False), time_decode(True)
        dec = dict(d_early)
        dec.update({"what": "one -phase test call of the reference = beam pass over max_decoder_l = 50 steps + gold pass (model.lua:376-627), on the parameters the "
                            "training loop STARTED from.  chars_per_s = the nominal B*50 decoder steps / time; executed_steps_* = the steps the greedy kernel "
                            "ran (a 32-row group leaves once every row has emitted EOS / PAD: the outputs are those of the 50-step loop); `no_early_exit` = the "
                            "same call with AOCR_NO_DEC_EARLY=1 (all 50 steps executed)",
                    "emitted_chars_per_s": world * nnz / (d_early["ms_per_call"] * 1e-3), "beam": BEAM, "no_early_exit": d_full})
        if rank == 0 and d_full.get("beam_ms"):
            us = d_full["beam_us_per_executed_step"]
            # what bounds a step of the whole-sequence greedy kernel: five dependent all-gathers inside a 32-CU group (out, h1, h2, attention context,
            # partial logits) -- nothing is streamed from HBM (weights in registers, context in L2).  Floor = 5 x the idle one-hop hand-off price of
            # MI355X_MICROARCH.md ("handoff-1to1", 0.8-1.0 us).  The launch chain instead: 6 dependent launches per step.
            floor = 5 * 1.0 if cluster else 6 * 1.45
            bound = "exchange-latency" if cluster else "launch-latency"
            # long strips (C5: T = 1785): a step's attention streams the bf16 context of every image once (the k hypotheses of an image share the rows:
            # attn_bf16_beam_kernel) -- past the 256 MB Infinity Cache that is an HBM stream per step, and the larger floor
            ctx_bytes = float(B) * T * Hd * 2
            if not cluster and ctx_bytes > 256e6 and ctx_bytes / 8e12 * 1e6 > floor:
                floor, bound = ctx_bytes / 8e12 * 1e6, "hbm (context stream of the attention step)"
            dec["decode_roofline"] = {"bound": bound, "unit": "us per executed decoder step (all rows of the batch advance one step)",
                                      "achieved": us, "floor": floor, "frac": floor / us if us > 0 else None, "launches_per_step": 1.0 / 50 if cluster else 6,
                                      "note": "floor: MI355X_MICROARCH.md price list (handoff-1to1 idle 0.8-1.0 us per dependent hop; boundary 1.45 us per dependent launch; "
                                              "HBM 8 TB/s for a context that does not fit the Infinity Cache); measured with every step executed (AOCR_NO_DEC_EARLY=1): beam_ms / 50"}
        m.params.copy_(trained[0]); m.bn_state.copy_(trained[1])
    if args.decode_steps > 0:
        # the same step under -use_dictionary: a synthetic 90 k-word lexicon as a device-resident flat trie (trained parameters: the
        # early exit is the point of this leg)
        if not args.no_secondary and BEAM == 1:
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
            eldd = (time.perf_counter() - t0) / args.decode_steps
            lab = m.decode_device(images, targets, targets_eval, 1, trie)[0]
            # the greedy kernel leaves its loop once every row of a 32-row group has emitted EOS / PAD: steps a group actually ran =
            # 1 + the last position at which any of its rows still emitted a token other than PAD / EOS (an upper bound: + the EOS step)
            live = ((lab != 1) & (lab != 3)).cpu().numpy()
            last = np.where(live.any(axis=1), live.shape[1] - np.argmax(live[:, ::-1], axis=1), 0)
            grp = [int(min(50, last[g:g + 32].max() + 2)) for g in range(0, B, 32)]
            executed = int(sum(min(32, B - g) * grp[g // 32] for g in range(0, B, 32)))
            dec_dict = {"chars_per_s": world * B * 50 / eldd, "executed_steps_per_s": world * executed / eldd, "executed_steps_per_call": executed,
                        "nominal_steps_per_call": B * 50, "ms_per_call": 1e3 * eldd,
                        "what": "chars_per_s divides the nominal B*50 decoder steps of a -phase test call by the time; the greedy kernel's early exit "
                                "(every row of a 32-row group at EOS / PAD) skips most of them under a dictionary: executed_steps_per_s is the "
                                "comparable rate", "words": len(words),
                        "trie_nodes": trie.n_nodes, "trie_bytes": int(trie.mask.nbytes + trie.base.nbytes + trie.child.nbytes)}

    # ---- secondary line (N = 1 only): BASELINE.json configs[1] = C2 in exact-fp32 MFMA mode (the 1e-4 logit-parity configuration)
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
        f2 = flops(w2["W"], w2["He"], w2["Le"], w2["Ld"], w2["L"])["total"]
        c2 = {"workload": "c2: " + w2["name"], "dtype": "f32", "steps": 10, "ms_per_step": 1e3 * e / 10, "value": w2["B"] * 10 / e,
              "unit": "image-lines/s", "step_mfma_frac": 3 * f2 * w2["B"] * 10 / e / 1e12 / PEAK["f32"]}
        m2.shutdown()

    # ---- beam-5 decode of the same lines (N = 1 only; model.lua:376-536 with -beam_size 5).  bf16, Hd = 512: the decoder chain kernel's BEAM variant
    # (dec_chain.hip: the 5 hypotheses of an image are rows of one chain, 6 images per 32-CU group, launches of <= 8 groups); the per-step launch chain
    # (AOCR_NO_DEC_CHAINS_BEAM=1: one cell / attention / project_select / state-gather launch set per step) is timed beside it
    beam5 = None
    if world == 1 and args.workload == "c3" and not args.no_secondary and args.decode_steps > 0:
        m5 = aocr.Model().create(dict(encoder_num_hidden=wl["He"], encoder_num_layers=wl["Le"], decoder_num_layers=wl["Ld"], input_feed=True,
                                      batch_size=B, max_img_w=W, max_decoder_l=50, max_beam=5, compute=wl["compute"], learning_rate=0.1, seed=910820))

        def time_beam():
            m5.decode_device(images, targets, targets_eval, 5); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.decode_steps):
                lab = m5.decode_device(images, targets, targets_eval, 5)[0]
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.decode_steps, lab.clone()
        e5, lab5 = time_beam()
        os.environ["AOCR_NO_DEC_CHAINS_BEAM"] = "1"
        e5c, lab5c = time_beam()
        del os.environ["AOCR_NO_DEC_CHAINS_BEAM"]
        beam5 = {"beam": 5, "chars_per_s": B * 50 / e5, "ms_per_call": 1e3 * e5,
                 "launch_chain_ms_per_call": 1e3 * e5c, "launch_chain_chars_per_s": B * 50 / e5c,
                 "labels_equal_to_launch_chain": float((lab5 == lab5c).float().mean().item()),
                 "what": "B*50 decoder steps per -phase test call at beam 5 (5 hypotheses per line advance per step), beam pass + gold pass; every step "
                         "executed (a finished hypothesis can still be overtaken: no early exit at beam > 1)"}
        m5.shutdown()

    # ---- data path, HBM-bound
    dp = None
    if world == 1 and not args.no_secondary:
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
              "ms_per_batch": msd, "bound": "hbm", "achieved_GBps": byts / (msd * 1e-3) / 1e9, "peak_GBps": HBM_PEAK_GBPS,
              "frac": byts / (msd * 1e-3) / 1e9 / HBM_PEAK_GBPS}

    out = None
    hbm_kernels = None
    if rank == 0 and wl["compute"] == "bf16" and world == 1 and not wl.get("widths"):
        # ---- the BANDWIDTH-bound kernels ("report HBM GB/s for those kernels separately"): each replayed by the library on the
        # buffers of a training step with the arguments the step passes (aocr_profile_kernel ids >= 2), HIP events on the model's stream;
        # algorithmic bytes = what the kernel must read and write once.  `frac` is against the 8 TB/s spec; ~6.3 TB/s is what a float4 copy
        # reaches on this chip (MI355X_MICROARCH.md), so 0.79 is the practical ceiling.  PMC cross-check: profiles/r05_hbm_pmc.txt.
        step(); torch.cuda.synchronize()
        names = {2: ("conv1_fwd_kernel", "normalise + conv1 + ReLU + 2x2 pool: fp32 image in, pooled bf16 map out"),
                 3: ("conv1_bwd_pk_kernel", "conv1 filter gradient: image + fp32 d(pooled map) in; VALU-bound (~110 instructions per window and lane), not byte-bound"),
                 4: ("bn_fwd_finalize_kernel + bn_apply_relu_kernel", "conv5 BatchNorm + ReLU: fp32 map in, bf16 out (sums from the conv epilogue)"),
                 5: ("bn_partial4_kernel<1> + bn_bwd_finalize_kernel + bn_bwd_apply_kernel", "conv5 BatchNorm backward: sums pass (10 B / element) + apply pass (12 B / element)"),
                 6: ("unpool8_kernel", "conv6 (2,1) un-pool + ReLU backward: fp32 d(pooled), arg-max, bf16 mask in; bf16 gradient map out"),
                 7: ("attn_dctx_kernel", "d(context) over the L decoder steps: (L,B,T) weights and score gradients + (L,B,Hd) vectors in, (B,T,Hd) fp32 out; the L x re-reads are served by L2"),
                 8: ("splitk_reduce_kernel", "sum of conv6's 8 split-K filter-gradient slabs (fp32) into the gradient")}
        hbm_kernels = {}
        for kid, (kname, what) in names.items():
            try:
                ms_k, by = m.profile_kernel(kid, 20)
            except Exception as e_:                       # a replay that does not exist at this workload's shape (aocr_profile_kernel refuses it): say so, keep the line
                hbm_kernels[kname] = {"skipped": str(e_)[:200]}
                continue
            gbps = by / (ms_k * 1e-3) / 1e9
            hbm_kernels[kname] = {"us_per_launch": 1e3 * ms_k, "algorithmic_MB": by / 1e6, "GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS,
                                  "frac_of_6.3TBps_achievable": gbps / 6300.0, "what": what}
        step(); torch.cuda.synchronize()                  # the replays clobbered activations / gradients: leave the model in a defined state
    if rank == 0:
        bf16 = wl["compute"] == "bf16"
        ms, kfl = m.profile_kernel(0, 20)                # conv6 forward, HIP events on the model's stream
        ach = kfl / (ms * 1e-3) / 1e12
        best = {"bound": "mfma", "kernel": "conv6 forward implicit GEMM (512->512 3x3 + ReLU + pool), gemm_halo4_bf16_kernel", "achieved": ach,
                "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "ms_per_launch": ms}
        # the DOMINANT kernel of the step by time share: the split-K filter gradient (conv_wgrad_dma_kernel: conv4 + conv5 + conv6, 12-13 % of
        # the kernel time, profiles/*_kernel_stats.csv); timed on the conv6 launch exactly as backward_all makes it (kernel + the sum of its slabs)
        msw, kflw = m.profile_kernel(1, 20)
        achw = kflw / (msw * 1e-3) / 1e12
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "r06_wgrad_pmc.json")
        if bf16 and args.workload == "c3" and world == 1 and args.scaling == "weak" and os.path.exists(pmc):
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
            traffic_source = ("profiles/r06_wgrad_pmc.json: rocprofv3 --pmc passes of this command on the tagged conv6 filter-gradient launch "
                              "(tools/pmc_traffic.py); PMC counters cannot be read from inside the timed process, so this field is NOT measured in this run")
        step_frac = 3 * fl["total"] * lines_per_s / 1e12 / (peak * world)
        wg_share = (families or {}).get("conv_wgrad", {}).get("ms_per_step", 0.0) / max(1e-9, (families or {}).get("_sum_ms", 1.0))
        roof = {"bound": "mfma", "kernel": "conv6 filter gradient (512x4608 over 65536 pixels): conv_wgrad_halo_kernel + splitk_reduce -- the largest launch of the "
                                           "dominant kernel family of the step by time (the filter gradients)",
                "achieved": achw, "peak": peak, "unit": "TFLOP/s", "frac": achw / peak, "traffic": traffic, "traffic_source": traffic_source,
                "ms_per_launch": msw, "algorithmic_gflop_per_launch": kflw / 1e9, "family_share_of_step": wg_share,
                "step_frac": step_frac, "step_frac_what": "SURVEY.md 8(d): train GFLOP per image x image-lines/s / bf16 MFMA peak over the WHOLE step (north-star target 0.40)"}
        if bf16:
            # context for `peak` (not a replacement for it): what a K loop that does nothing but v_mfma_f32_32x32x16_bf16 on register-resident RANDOM bf16
            # operands sustains on this chip -- the shader clock falls from 2.4 to ~1.7 GHz under it (power); profiles/r03_gemm4w_ubench.txt, tools/ubench/gemm4w.hip
            mo, mo_src = None, os.path.join("profiles", "r06_gemm4w_steady.txt")
            try:                                          # tools/ubench/gemm4w after 5000 warm-up launches: the "MFMA on resident random fragments" rows, best of the file
                mo = max(float(l.split("TFLOP/s")[0].split()[-1]) for l in open(os.path.join(ROOT, mo_src)) if "MFMA on resident random fragments" in l)
            except Exception:
                pass
            for r_ in (roof, best):
                r_["mfma_only_sustained_TFLOPs"] = mo
                r_["frac_of_mfma_only_sustained"] = r_["achieved"] / mo if mo else None
                r_["mfma_only_source"] = mo_src + " (tools/ubench/gemm4w.hip, measured this round on the pool; a file constant, NOT measured in this run)"
        out = {
            "metric": "image-lines/sec (train step)", "value": lines_per_s, "unit": "image-lines/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {wl['name']}", "global_batch": global_B, "per_gpu_batch": B, "img": f"{IMG_H}x{W}",
                       "decoder_steps": L, "parallelism": f"dp{world}", "input_feed": True},
            "scaling_measured": world > 1, "rccl_ranks": rccl_ranks, "cluster_fallback": cluster_fallback,
            "exchange_provider": args.provider if world > 1 else None, "exchange_exposed_ms": exposed, "dropout": args.dropout,
            "step_ms_events": {"median": float(np.median(per_step)), "p10": float(np.percentile(per_step, 10)),
                               "p90": float(np.percentile(per_step, 90)), "n": int(args.steps)},
            "sustained": sustained,
            "train_gflop_per_image": 3 * fl["total"] / 1e9, "step_tflops": 3 * fl["total"] * lines_per_s / 1e12,
            "step_mfma_frac": 3 * fl["total"] * lines_per_s / 1e12 / (peak * world),
            "families": families,
            "decode_chars_per_s": dec["chars_per_s"] if dec else None, "decode": dec, "decode_dict": dec_dict, "decode_beam5": beam5,
            "replica_drift": replica_drift, "loss": loss_val, "secondary": c2, "data_path": dp,
            "roofline": roof, "roofline_best": best, "hbm_kernels": hbm_kernels,
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
