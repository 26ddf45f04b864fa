#!/bin/bash
# one training step of a bench workload as an ordered kernel list (name, grid, workgroup, us): tools/ktrace_step.sh <workload> [extra bench args]
# (run via gpurun) -> gpurun_out/ktrace_step_<workload>.txt
set -u
R=${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run through gpurun)}; wl=${1:?usage: ktrace_step.sh <workload> [bench args]}; shift; O=$R/gpurun_out/ktrace_step_raw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $wl --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 "$@" > $O/run.log 2>&1
cd $R
python3 - "$(ls $O/*/*kernel_trace.csv | head -1)" > gpurun_out/ktrace_step_$wl.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
# the last full step: from the last-but-one sgd_update_kernel to the last one
ends = [i for i, r in enumerate(rows) if "sgd_update_kernel" in r["Kernel_Name"]]
# bench runs profile/family passes after the timed steps: take the step that ends at the (warmup + steps)-th update
k = min(len(ends) - 1, 5)
lo, hi = ends[k - 1] + 1, ends[k]
gk = [c for c in rows[0].keys() if "Grid" in c]; wk = [c for c in rows[0].keys() if "Workgroup_Size" in c or "Workgroup" in c]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = "x".join(r[c] for c in gk[:3]); w = "x".join(r[c] for c in wk[:3])
    q = r.get("Stream_Id") or r.get("Queue_Id") or "?"       # which HIP stream (hardware queue) the launch went to: the side-stream overlaps are read off this column
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  -> {(e - t0) / 1e3:8.1f}  q{q:>2s}  grid {g:>16s} wg {w:>10s}  {r['Kernel_Name'][:150]}")
print(f"step: {(int(rows[hi]['End_Timestamp']) - t0) / 1e3:.1f} us, {hi - lo + 1} launches")
PY
rm -rf $O
