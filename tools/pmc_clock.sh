#!/bin/bash
# Clock and MFMA-busy fraction per kernel of one bench workload (run via gpurun): tools/pmc_clock.sh <workload> [extra bench args]
# one --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE) with --kernel-trace; clock = GRBM_GUI_ACTIVE / (8 XCDs x dispatch duration) -> gpurun_out/pmc_clock_<workload>.txt
R=$GRAFT_REPO_ROOT; wl=$1; shift; O=$R/gpurun_out/pmc_clock_raw; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --decode-steps 0 --sustain-seconds 0 "$@" > $O/run.log 2>&1
cd $R
python3 - "$(ls $O/*/*counter_collection.csv | head -1)" "$(ls $O/*/*kernel_trace.csv | head -1)" > gpurun_out/pmc_clock_$wl.txt <<'PY'
import csv, sys, collections
dur = {}
for r in csv.DictReader(open(sys.argv[2])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    cnt[r["Dispatch_Id"]]["_grid"] = r.get("Grid_Size", "")
acc = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for d, c in cnt.items():
    if d not in dur or "GRBM_GUI_ACTIVE" not in c: continue
    ns, name = dur[d]
    a = acc[(name[:100], c["_grid"])]; a[0] += 1; a[1] += ns; a[2] += c["GRBM_GUI_ACTIVE"]; a[3] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
print("# launches, mean us (counter pass: serialised dispatches), clock GHz = GRBM_GUI_ACTIVE / (8 XCDs x duration), MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE), grid, kernel")
for (name, grid), a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{a[0]:4d}  {a[1]/a[0]/1e3:8.1f} us  {a[2]/a[1]/8:5.2f} GHz  mfma_busy {a[3]/(128.0*a[2]):5.3f}  grid {grid:>9s}  {name}")
PY
rm -rf $O
