"""per-kernel averages of every counter collected by tools/pmc_diag.sh: python tools/pmc_diag_summary.py <dir> [name-filter]"""
import csv, glob, os, sys, collections
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "igemm")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if flt not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print(k, "launches", max(cnt[k].values()))
    a = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    for c in sorted(a): print("   %-40s %14.4g" % (c, a[c]))
    w = a.get("SQ_WAVE_CYCLES")
    if w:
        print("   -> of wave cycles: wait_any %.2f  wait_inst %.2f  active %.2f | active valu %.2f vmem %.2f lds %.2f" % tuple(a.get(c, 0) / w for c in
              ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS")))
    if a.get("TCC_REQ_sum"): print("   -> L2 hit rate %.3f" % (a.get("TCC_HIT_sum", 0) / (a.get("TCC_HIT_sum", 0) + a.get("TCC_MISS_sum", 1e-9))))
    if a.get("TCP_TOTAL_CACHE_ACCESSES_sum"): print("   -> L1 miss share (TCC read req / L1 accesses) %.3f" % (a.get("TCP_TCC_READ_REQ_sum", 0) / a["TCP_TOTAL_CACHE_ACCESSES_sum"]))
    if a.get("SQ_LDS_IDX_ACTIVE"): print("   -> LDS bank-conflict share %.3f" % (a.get("SQ_LDS_BANK_CONFLICT", 0) / a["SQ_LDS_IDX_ACTIVE"]))
