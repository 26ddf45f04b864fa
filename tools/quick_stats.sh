#!/bin/bash
# Kernel-time summary of a short bench run on the GPU box: bash tools/quick_stats.sh [extra bench args] -> gpurun_out/qs/kernel_stats.csv + top lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qs; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 "$@" > $O/bench.log 2>&1
cd $R
cp $(ls $O/raw/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/raw
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/qs/kernel_stats.csv"))))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{float(r["TotalDurationNs"]) / 1e3 / 13:9.1f} us/step  calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:90]}')
print("total", tot / 1e6 / 13, "ms/step over 13 steps")
PY
tail -c 300 $O/bench.log
