#!/bin/bash
# Kernel-time summary of 20 decode calls (beam 1 + gold pass) at the C3 shape: bash tools/decode_stats.sh -> gpurun_out/ds/kernel_stats.csv + top lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ds; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python $R/tools/decode_prof.py 20 > $O/run.log 2>&1
cd $R
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/raw
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/ds/kernel_stats.csv"))))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"]) / 1e3 / 20:9.1f} us/call  calls/call {int(r["Calls"]) / 20:6.2f}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:100]}')
print("total", tot / 1e6 / 20, "ms of kernel time per decode call")
PY
