"""HBM traffic per launch of the conv6-forward kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KB;
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced read stream -> doubled."""
import csv
import sys


def conv6(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "gemm_halo_bf16_kernel<aocr::EpConv, 1, 256, 256, 1>" in r["Kernel_Name"]]   # the tagged launches of aocr_profile_kernel
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    vals = [float(r["Counter_Value"]) for r in rows[-20:]]        # the 20 timed launches of aocr_profile_kernel
    return sum(vals) / len(vals), len(rows)


f, nf = conv6(sys.argv[1], "FETCH_SIZE")
w, nw = conv6(sys.argv[2], "WRITE_SIZE")
fetch_bytes = 2.0 * f * 1024.0
write_bytes = w * 1024.0
print(f"conv6 fwd launches seen: {nf}/{nw}; FETCH_SIZE {f:.0f} KB (x2 corrected -> {fetch_bytes/1e6:.1f} MB), WRITE_SIZE {w:.0f} KB ({write_bytes/1e6:.1f} MB)")
print(f"traffic per launch: {(fetch_bytes + write_bytes)/1e6:.1f} MB")
