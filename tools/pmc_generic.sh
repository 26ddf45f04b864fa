#!/bin/bash
# per-kernel mean of arbitrary PMC counters over the default workload (run via gpurun): tools/pmc_generic.sh <tag> <counter...> -> gpurun_out/pmc/<tag>.txt
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/pmc; mkdir -p $O; rm -rf $O/raw_$tag
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/raw_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 1 --no-secondary --sustain-seconds 0 > $O/run_$tag.log 2>&1
cd $R
f=$(ls $O/raw_$tag/*/*counter_collection.csv | head -1); kt=$(ls $O/raw_$tag/*/*kernel_trace.csv | head -1)
python - "$f" "$kt" "$@" > $O/$tag.txt <<'PY'
import csv, sys, collections
names = sys.argv[3:] + ["duration_us_in_this_run"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(sys.argv[2])):                        # durations of the SAME (counter-collecting, serialised) run
    acc[r["Kernel_Name"]]["duration_us_in_this_run"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in acc.items():
    n = max(len(v) for v in c.values())
    key = c.get("GRBM_GUI_ACTIVE") or c.get(names[0]) or [0]
    rows.append((sum(key), k, n, {x: (sum(c[x]) / len(c[x]) if x in c else float("nan")) for x in names}))
rows.sort(key=lambda r: -r[0])
print("# rocprofv3 --pmc " + " ".join(names) + " --kernel-trace -- python bench.py --steps 2 --warmup 1 (tools/pmc_generic.sh); per kernel: launches, mean counter values per launch")
for tot, k, n, m in rows[:26]:
    print(f"{n:4d} " + " ".join(f"{x}={m[x]:.4g}" for x in names) + "  " + k[:100])
PY
rm -rf $O/raw_$tag
