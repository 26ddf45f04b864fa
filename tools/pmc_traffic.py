"""HBM-side traffic per launch of the two kernels bench.py's roofline fields name, from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE) of the bench command: the tagged launches of aocr_profile_kernel -- id 0 = conv6 forward (gemm_halo4_bf16_kernel<EpConv, 1, TAG 1>; under AOCR_HALO8=1 the 8-wave gemm_halo_bf16_kernel<..., 1>),
id 1 = conv6 filter gradient (round 4: conv_wgrad_halo_kernel<1>; round 3: conv_wgrad_dma_kernel<EpStore, 256>; + the splitk_reduce_kernel launch behind each of them).
Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KB; FETCH_SIZE reports
exactly half the bytes of a wide (16 B/lane) coalesced read stream -> doubled.  Writes JSON next to the text summary when asked:
    python tools/pmc_traffic.py fetch.csv write.csv [out_dir]"""
import csv
import json
import os
import sys

FWD = "gemm_halo4_bf16_kernel<aocr::EpConv, 1, 1>"
WG = "conv_wgrad_halo_kernel<1,"          # round 4 (round 3: "conv_wgrad_dma_kernel<aocr::EpStore, 256>")
RED = "splitk_reduce_kernel"


def rows_of(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def per_launch(rows, name, last=20, follow=None):
    """mean counter value over the last `last` launches of kernel `name`; follow: also the launch of kernel `follow` right behind each."""
    idx = [i for i, r in enumerate(rows) if name in r["Kernel_Name"]]
    idx = idx[-last:]
    main = [float(rows[i]["Counter_Value"]) for i in idx]
    extra = []
    if follow:
        for i in idx:
            if i + 1 < len(rows) and follow in rows[i + 1]["Kernel_Name"]:
                extra.append(float(rows[i + 1]["Counter_Value"]))
    return (sum(main) / max(1, len(main)), len(idx), (sum(extra) / len(extra)) if extra else 0.0)


def main():
    fr, wr = rows_of(sys.argv[1], "FETCH_SIZE"), rows_of(sys.argv[2], "WRITE_SIZE")
    out_dir = sys.argv[3] if len(sys.argv) > 3 else None
    f, nf, _ = per_launch(fr, FWD); w, nw, _ = per_launch(wr, FWD)
    fb, wb = 2.0 * f * 1024.0, w * 1024.0
    print(f"conv6 fwd launches seen: {nf}/{nw}; FETCH_SIZE {f:.0f} KB (x2 corrected -> {fb/1e6:.1f} MB), WRITE_SIZE {w:.0f} KB ({wb/1e6:.1f} MB)")
    print(f"traffic per launch: {(fb + wb)/1e6:.1f} MB")
    gf, ngf, rf = per_launch(fr, WG, follow=RED); gw, ngw, rw = per_launch(wr, WG, follow=RED)
    gfb, gwb, rfb, rwb = 2.0 * gf * 1024.0, gw * 1024.0, 2.0 * rf * 1024.0, rw * 1024.0
    print(f"conv6 filter gradient launches seen: {ngf}/{ngw}; kernel reads {gfb/1e6:.1f} MB writes {gwb/1e6:.1f} MB; slab sum reads {rfb/1e6:.1f} MB writes {rwb/1e6:.1f} MB")
    print(f"traffic per launch (kernel + slab sum): {(gfb + gwb + rfb + rwb)/1e6:.1f} MB")
    if out_dir:
        cmd = ("rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
               "--decode-steps 0 --no-secondary (two passes; tools/collect_profiles.sh), tools/pmc_traffic.py over the tagged launches of aocr_profile_kernel")
        note = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (16 B/lane coalesced reads report half); Infinity-Cache hits are counted, so this is "
                "L2-miss traffic")
        json.dump({"kernel": FWD + " (conv6 forward, C3 shape B=256 W=256)", "command": cmd, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB": w,
                   "fetch_bytes_corrected_x2": fb, "write_bytes": wb, "traffic_bytes_per_launch": fb + wb,
                   "algorithmic_bytes_per_launch": {"read_A5_bf16": 67108864, "read_w_bf16": 4718592, "write_idx_u8": 16777216, "write_A6_bf16": 33554432},
                   "note": note}, open(os.path.join(out_dir, "conv6_fwd_pmc.json"), "w"), indent=1)
        json.dump({"kernel": "conv_wgrad_halo_kernel<1, 4> + splitk_reduce_kernel (conv6 filter gradient, C3 shape: 512 x 4608 over 65536 pixels, split-K 8 over 256 workgroups of 256 x 288)",
                   "command": cmd, "kernel_fetch_bytes_corrected_x2": gfb, "kernel_write_bytes": gwb, "reduce_fetch_bytes_corrected_x2": rfb,
                   "reduce_write_bytes": rwb, "traffic_bytes_per_launch": gfb + gwb + rfb + rwb,
                   "algorithmic_bytes_per_launch": {"read_A5_bf16": 67108864, "read_dY6_bf16": 67108864, "write_dW_f32": 9437184},
                   "split_k_bytes": {"slab_writes_f32": 8 * 9437184, "slab_reads_f32": 8 * 9437184, "dW_read_modify_write": 2 * 9437184},
                   "note": note + ".  The split-K partial tiles (7 k-ranges x 9.4 MB) are written once as plain stores and read once by the slab sum; "
                                  "in round 2 they were 66 MB of fp32 atomics executed at the memory side"},
                  open(os.path.join(out_dir, "wgrad_pmc.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
