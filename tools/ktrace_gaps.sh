#!/bin/bash
# kernel-trace of a short bench run; prints, for selected kernels, the mean duration and the mean idle gap before / after them (run via gpurun)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 > $O/run.log 2>&1
cd $R
python3 - "$(ls $O/*/*kernel_trace.csv | head -1)" "$@" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
keys = sys.argv[2:] or ["wgrad", "splitk"]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if not any(k in n for k in keys): continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    before = s - int(rows[i - 1]["End_Timestamp"]) if i else 0
    after = int(rows[i + 1]["Start_Timestamp"]) - e if i + 1 < len(rows) else 0
    a = acc[n[:90]]; a[0] += 1; a[1] += (e - s) / 1e3; a[2] += before / 1e3; a[3] += after / 1e3
for n, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[0]:5d} launches  mean {a[1]/a[0]:8.1f} us  gap before {a[2]/a[0]:7.1f} us  gap after {a[3]/a[0]:7.1f} us  {n}")
PY
rm -rf $O
