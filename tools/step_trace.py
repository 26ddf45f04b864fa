"""Per-launch listing of one train step from a rocprofv3 --kernel-trace CSV (last full step between sgd_update kernels)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "sgd_update" in n]
s, e = ends[-2] + 1, ends[-1] + 1
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm_lds"
agg = {}
for r in rows[s:e]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"]
    key = n.split("(")[0].replace("aocr::", "").replace("void ", "")
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
    if flt in n:
        short = n.split("<")[1].split(">")[0].replace("aocr::", "") if "<" in n else n.split("(")[0]
        print(f"{d:9.1f} us  grid {r['Grid_Size_X']:>8} z{r['Grid_Size_Z']:>4} wg {r['Workgroup_Size_X']:>4}  {short}")
t0 = int(rows[s]["Start_Timestamp"]); t1 = int(rows[e - 1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[s:e])
print("step wall us", (t1 - t0) / 1e3, "busy us", busy / 1e3, "launches", e - s)
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t:10.1f} us {c:5d}x  {k[:150]}")
