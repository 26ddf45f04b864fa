#!/bin/bash
# kernel list of the END of a bench run of BASELINE config 5 (the greedy decode calls come last): tools/debug/trace_decode_c5.sh [batch] -> gpurun_out/dec_trace_c5.txt
set -u
R=${GRAFT_REPO_ROOT:?run through gpurun}; B=${1:-256}; O=$R/gpurun_out/dec_trace_raw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload c5 --batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --decode-steps 2 --sustain-seconds 0 > $O/run.log 2>&1
cd $R
python3 - "$(ls $O/*/*kernel_trace.csv | head -1)" > gpurun_out/dec_trace_c5.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-260:]
t0 = int(tail[0]["Start_Timestamp"])
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:120]}")
PY
rm -rf $O
