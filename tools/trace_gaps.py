"""idle time of the GPU inside the timed steps: union of kernel intervals from a rocprofv3 --kernel-trace csv
usage: python tools/trace_gaps.py <kernel_trace.csv> [n_last_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# steps are delimited by adam_kernel launches
adam = [i for i, x in enumerate(iv) if x[2].startswith("adam_kernel")]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 5
print("adam launches:", len(adam))
for a, b in list(zip(adam[:-1], adam[1:]))[-nlast:]:
    seg = iv[a + 1:b + 1]
    t0, t1 = iv[a][1], iv[b][1]
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, n in seg:
        if cur_e is None:
            cur_s, cur_e = s, e
            if s - t0 > 0: gaps.append((s - t0, "<step start>", n))
        elif s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, prev, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        prev = n
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    print("step %.3f ms  busy %.3f ms  idle %.3f ms  kernels %d  sum-of-kernels %.3f ms" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(seg), sum(e - s for s, e, _ in seg) / 1e6))
    for g, p, n in gaps[:8]:
        print("   gap %.1f us  after %s  before %s" % (g / 1e3, p[:50], n[:50]))
