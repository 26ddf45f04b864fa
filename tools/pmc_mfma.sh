#!/bin/bash
# MFMA-busy / wait counters per kernel family of the default workload (run via gpurun): gpurun_out/pmc/mfma_busy.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/raw -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 1 --no-secondary --sustain-seconds 0 > $O/run.log 2>&1
cd $R
f=$(ls $O/raw/*/*counter_collection.csv | head -1)
python - "$f" > $O/mfma_busy.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c: continue
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]); mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    rows.append((g * len(c["GRBM_GUI_ACTIVE"]), k, len(c["GRBM_GUI_ACTIVE"]), g, mf))
rows.sort(reverse=True)
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -- python bench.py --steps 2 --warmup 1 (tools/pmc_mfma.sh)")
print("# per kernel: launches, mean GRBM_GUI_ACTIVE cycles, mean SQ_VALU_MFMA_BUSY_CYCLES, MFMA-busy fraction = busy / (128 x active): the counter comes back averaged over the 8 XCDs, 32 CUs x 4 SIMDs each (the same normalisation gives the 45 % / 31 % of DESIGN.md round 1)")
for tot, k, n, g, mf in rows[:24]:
    print(f"{n:4d}  active {g:12.0f}  mfma_busy {mf:14.0f}  mfma_busy_frac {mf / (128.0 * g):6.3f}  {k[:110]}")
PY
rm -rf $O/raw
cat $O/mfma_busy.txt | head -30
