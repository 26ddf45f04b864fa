"""HBM-side traffic per launch of the bandwidth-bound kernels bench.py lists under `hbm_kernels`, from the two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) of the bench command (tools/round_profiles.sh).  Counters are in KB; FETCH_SIZE is doubled (gfx950 reports half the
bytes of a 16-B-per-lane coalesced read stream: /opt/skills/guides/MI355X_MICROARCH.md, HBM section) -- for the kernels here whose loads are
narrower (conv1: 4-byte image loads) the doubling is an upper bound; Infinity-Cache hits are counted, so this is L2-miss traffic.
    python tools/pmc_hbm_kernels.py fetch_counter_collection.csv write_counter_collection.csv kernel_trace.csv"""
import collections
import csv
import sys

KERNELS = ["conv1_fwd_kernel", "conv1_bwd_pk_kernel", "bn_partial4_kernelILi1", "bn_partial4_kernelILi0", "bn_apply_relu_kernel", "bn_bwd_apply_kernel",
           "unpool8_kernel", "attn_dctx_kernel", "splitk_reduce_kernel", "colsum_jobs_kernel", "sgd_update_kernel", "shadow_jobs_kernel"]


def mean_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = mean_by_kernel(sys.argv[1], "FETCH_SIZE"), mean_by_kernel(sys.argv[2], "WRITE_SIZE")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[3])):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0 (two passes)")
print("# per kernel symbol, mean over ALL its launches of the run (every shape it is launched with: the three BatchNorm layers, the three un-pool layers, ...):")
print("# launches | fetch MB (x2 corrected) | write MB | total MB | us in the counter run | GB/s in the counter run")
for key in KERNELS:
    for name in fetch:
        if key in name:
            f = 2.0 * 1024.0 * sum(fetch[name]) / len(fetch[name]); w = 1024.0 * sum(write.get(name, [0.0])) / max(1, len(write.get(name, [0.0])))
            d = sum(dur[name]) / max(1, len(dur[name]))
            print(f"{len(fetch[name]):5d} | {f / 1e6:9.2f} | {w / 1e6:9.2f} | {(f + w) / 1e6:9.2f} | {d:8.1f} | {(f + w) / max(d, 1e-9) / 1e3:8.0f}  {name[:110]}")
