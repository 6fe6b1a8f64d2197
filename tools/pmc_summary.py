"""Summarise the rocprofv3 --pmc passes written by tools/pmc_collect.sh.

Per kernel name: launches, mean duration, HBM-side read/write bytes per launch and the MFMA-busy share.
  * FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1 KB (TCC_EA0_RDREQ * 64 B / 1024);
    MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced
    reads -> the "read x2" column is the corrected figure for streaming kernels; WRITE_SIZE is exact for
    16-B-per-lane stores and float atomics.  Narrow (4-B-per-lane gather) reads are uncalibrated, so for the
    im2col kernels the truth lies between "read" and "read x2".
  * mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * kernel cycles), the share of SIMD-cycles with the
    matrix pipe busy.  Calibration on this box: SQ_VALU_MFMA_BUSY_CYCLES is exactly 64 per v_mfma_f32_32x32x2_f32
    (summed over the chip) and GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = GRBM_GUI_ACTIVE / 8,
    which puts the clock at ~2.23 GHz under this load, against the 2.4 GHz behind the 157.3 TFLOP/s peak).
  * calibration of the read correction in OUR access patterns: adam_kernel reads 4 x 358 MB and act_bwd_kernel reads
    two tensors per tensor written; both report exactly half of that in FETCH_SIZE -> reads are doubled ("read_x2").

usage: python tools/pmc_summary.py gpurun_out/pmc_r1c [out.csv]
"""
import csv, collections, os, sys

root = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
CUS = 256


def load(tag):
    rows = list(csv.DictReader(open(os.path.join(root, tag, "pmc_counter_collection.csv"))))
    per = collections.defaultdict(dict)   # dispatch -> counter -> value
    meta = {}
    for r in rows:
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] = float(r["Counter_Value"])
        meta[d] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return per, meta


def short(n):
    n = n.replace("void ", "")
    return n.split("(")[0]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES"):
    per, meta = load(tag)
    for d, c in per.items():
        k = short(meta[d][0])
        a = agg[k]
        a["n_" + tag] += 1
        if tag == "SQ_VALU_MFMA_BUSY_CYCLES":
            a["dur_ns"] += meta[d][1]      # durations from the lightest pass
        for name, v in c.items():
            a[name] += v

rows = []
for k, a in agg.items():
    n = a["n_SQ_VALU_MFMA_BUSY_CYCLES"] or 1
    nf, nw = a["n_FETCH_SIZE"] or 1, a["n_WRITE_SIZE"] or 1
    rd = a["FETCH_SIZE"] * 1024 / nf
    wr = a["WRITE_SIZE"] * 1024 / nw
    busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * CUS * a["GRBM_GUI_ACTIVE"] / 8) if a["GRBM_GUI_ACTIVE"] else 0.0
    ghz = a["GRBM_GUI_ACTIVE"] / 8 / a["dur_ns"] if a["dur_ns"] else 0.0
    us = a["dur_ns"] / n / 1e3
    rows.append(dict(kernel=k, launches=int(n), avg_us=round(us, 1), read_MB=round(rd / 1e6, 2), read_x2_MB=round(2 * rd / 1e6, 2),
                     write_MB=round(wr / 1e6, 2), hbm_GBps_x2=round((2 * rd + wr) / (us * 1e-6) / 1e9, 1) if us else 0,
                     mfma_busy=round(busy, 3), clock_GHz=round(ghz, 2), total_ms=round(a["dur_ns"] / 1e6, 3)))
rows.sort(key=lambda r: -r["total_ms"])
cols = list(rows[0].keys())
if out:
    with open(out, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=cols)
        w.writeheader()
        w.writerows(rows)
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_tag
    rec = {r["kernel"]: {"hbm_bytes_per_launch": int((r["read_x2_MB"] + r["write_MB"]) * 1e6), "mfma_busy": r["mfma_busy"],
                         "avg_us": r["avg_us"], "launches": r["launches"]} for r in rows}
    # which kernels these counters belong to: bench.py only quotes them for a library built from the same sources
    rec["_kernel_source_tag"] = kernel_source_tag()
    with open(os.path.splitext(out)[0] + ".json", "w") as f:   # what bench.py's roofline.traffic reads
        json.dump(rec, f, indent=1, sort_keys=True)
print(" ".join("%-12s" % c if c != "kernel" else "%-44s" % c for c in cols))
for r in rows[:40]:
    print(" ".join(("%-44s" % str(r[c])[:44]) if c == "kernel" else "%-12s" % r[c] for c in cols))
