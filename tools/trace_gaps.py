"""Idle time between the kernels of a training step, from a rocprofv3 --kernel-trace CSV (columns Start_Timestamp / End_Timestamp / Kernel_Name):
steps are delimited by conv1_fwd_kernel launches; per step: wall span, sum of kernel durations on the busiest-stream view (union of intervals),
idle = span - union, and the number of kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "conv1_fwd_kernel" in e[2]]
out = []
for a, b in zip(starts[:-1], starts[1:]):
    seg = ev[a:b]
    if not any("sgd_update" in e[2] or "adadelta" in e[2] for e in seg): continue      # a decode call, not a train step
    span = seg[-1][1] - seg[0][0]
    union = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s <= cur_e: cur_e = max(cur_e, e)
        else: union += cur_e - cur_s; cur_s, cur_e = s, e
    union += cur_e - cur_s
    gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1]), seg[i][2][:60], seg[i + 1][2][:60]) for i in range(len(seg) - 1)), reverse=True)
    out.append((span, union, len(seg), gaps[:6]))
out = out[len(out) // 2:]                                                           # the later (warm) steps
n = len(out)
print(f"{n} train steps: span {sum(o[0] for o in out)/n/1e3:.1f} us, busy (union of kernel intervals) {sum(o[1] for o in out)/n/1e3:.1f} us, idle {sum(o[0]-o[1] for o in out)/n/1e3:.1f} us, kernels per step {sum(o[2] for o in out)/n:.0f}")
for g in out[-1][3]: print(f"  gap {g[0]/1e3:7.1f} us after {g[1]} -> {g[2]}")
