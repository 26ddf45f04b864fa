#!/bin/bash
# per-ablation duration of the whole-sequence encoder forward kernel (rocprofv3 kernel trace of 2 bench steps)
cd /tmp && export TMPDIR=/tmp
for abl in 0 1 2 4 3 7; do
  rm -rf /tmp/abl$abl
  AOCR_SEQ_ABL=$abl rocprofv3 --kernel-trace --output-format csv -d /tmp/abl$abl -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 > /dev/null 2>&1
  f=$(ls /tmp/abl$abl/*/*kernel_trace.csv | head -1)
  python - "$f" $abl <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "enc_seq_fwd" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print(f"abl {sys.argv[2]}: enc_seq_fwd {sum(d)/len(d):7.1f} us over {len(d)} launches")
PY
done
