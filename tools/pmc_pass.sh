#!/bin/bash
# One rocprofv3 --pmc pass over a short bench run, summarised per kernel (run via gpurun): tools/pmc_pass.sh <workload> <tag> "<COUNTER ...>" [extra bench args]
# -> gpurun_out/pmc_<tag>_<workload>.txt (mean of every counter per kernel symbol + grid, 25 kernels with the largest total duration)
R=$GRAFT_REPO_ROOT; wl=$1; tag=$2; ctr=$3; shift 3; O=$R/gpurun_out/pmc_pass_raw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 "$@" > $O/run.log 2>&1
cd $R
python3 - "$(ls $O/*/*counter_collection.csv | head -1)" "$(ls $O/*/*kernel_trace.csv | head -1)" > gpurun_out/pmc_${tag}_$wl.txt <<'PY'
import csv, sys, collections
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cnt = collections.defaultdict(dict); names = []
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"]); cnt[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "")
    if r["Counter_Name"] not in names: names.append(r["Counter_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d, c in cnt.items():
    if d not in dur: continue
    a = acc[(dur[d][1][:90], c["_grid"])]; a["_n"] += 1; a["_ns"] += dur[d][0]
    for n in names: a[n] += c.get(n, 0.0)
print("# per kernel: launches, mean us, then the MEAN per launch of: " + " ".join(names))
for (name, grid), a in sorted(acc.items(), key=lambda kv: -kv[1]["_ns"])[:25]:
    print(f"{int(a['_n']):4d} {a['_ns']/a['_n']/1e3:8.1f} us " + " ".join(f"{a[n]/a['_n']:14.0f}" for n in names) + f"  grid {grid:>8s} {name}")
PY
rm -rf $O
