"""Print ms_per_step and the per-family times of bench.py JSON lines: python tools/fam.py a.json b.json ..."""
import json, sys
rows = []
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    fam = d.get("families") or {}
    rows.append((f, d["ms_per_step"], {k: v["ms_per_step"] for k, v in fam.items() if isinstance(v, dict)}))
keys = sorted({k for r in rows for k in r[2]})
print("%-28s %8s " % ("file", "ms/step") + " ".join("%11s" % k[:11] for k in keys))
for f, ms, fam in rows:
    print("%-28s %8.3f " % (f[-28:], ms) + " ".join("%11.3f" % fam.get(k, 0.0) for k in keys))
