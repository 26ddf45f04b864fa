"""MFMA-busy fraction per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES pass of the bench command
(tools/round_profiles.sh): python tools/pmc_mfma_summary.py counter_collection.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in acc.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    g = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]); mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    rows.append((g * len(c["GRBM_GUI_ACTIVE"]), k, len(c["GRBM_GUI_ACTIVE"]), g, mf))
rows.sort(reverse=True)
print("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-steps 0 --no-secondary --sustain-seconds 0")
print("# per kernel SYMBOL OF THE SHIPPED DISPATCH: launches, mean GRBM_GUI_ACTIVE cycles, mean SQ_VALU_MFMA_BUSY_CYCLES, MFMA-busy fraction = busy / (128 x active)")
print("# (the counter comes back summed over the 8 XCDs' 32 CUs x 4 SIMDs; the same normalisation as profiles/r02_mfma_busy.txt, r03_mfma_busy.txt)")
for tot, k, n, g, mf in rows[:30]:
    print(f"{n:4d}  active {g:12.0f}  mfma_busy {mf:14.0f}  mfma_busy_frac {mf / (128.0 * g):6.3f}  {k[:130]}")
