"""Timeline reading of a rocprofv3 --kernel-trace CSV of bench.py (normal multi-stream mode): per step (Adam to Adam) the wall
time, the union of kernel-busy time, the idle gaps, the mean number of kernels in flight, and the time during which exactly one
kernel with a small grid (< 512 workgroups: at most two per CU) is alone on the chip -- with the kernels that account for it.
(The profiler adds host time per launch: a host-bound configuration shows more idle here than it has unprofiled.)
usage: python tools/trace_timeline.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    ev.append((s, e, r["Kernel_Name"], grid // max(wg, 1), r.get("Queue_Id", "")))
ev.sort()
adam = [i for i, x in enumerate(ev) if x[2].startswith("adam_kernel")]
print("kernels %d, steps %d, queues %s" % (len(ev), len(adam), sorted(set(x[4] for x in ev))))


def short(n):
    return n.replace("void ", "").split("(")[0][:44]


pairs = list(zip(adam[:-1], adam[1:]))[-4:]
for a, b in pairs:
    seg = ev[a + 1:b + 1]
    t0, t1 = ev[a][1], ev[b][1]
    pts = sorted([(s, 1, wgs, nm) for s, e, nm, wgs, q in seg] + [(e, -1, wgs, nm) for s, e, nm, wgs, q in seg])
    busy = idle = area = 0
    live, last, last_end = [], t0, None
    alone, gap_after = {}, {}
    for t, d, wgs, nm in pts:
        dt = t - last
        if dt > 0:
            if not live:
                idle += dt
                if last_end is not None:
                    gap_after[last_end] = gap_after.get(last_end, 0) + dt
            else:
                busy += dt
                area += dt * len(live)
                if len(live) == 1 and live[0][0] < 512:
                    alone[live[0][1]] = alone.get(live[0][1], 0) + dt
        last = t
        if d == 1:
            live.append((wgs, nm))
        else:
            live.remove((wgs, nm))
            last_end = nm
    tot = sum(e - s for s, e, *_ in seg)
    print("step: wall %.2f ms | busy (union) %.2f | idle %.2f | sum of kernel times %.2f | mean in flight %.2f | alone with < 512 workgroups %.2f ms | %d launches"
          % ((t1 - t0) / 1e6, busy / 1e6, idle / 1e6, tot / 1e6, area / max(busy, 1), sum(alone.values()) / 1e6, len(seg)))
    if (a, b) == pairs[-1]:
        print("   alone & small: " + ", ".join("%s %.0f us" % (short(k), v / 1e3) for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:16]))
        print("   idle after:    " + ", ".join("%s %.0f us" % (short(k), v / 1e3) for k, v in sorted(gap_after.items(), key=lambda kv: -kv[1])[:12]))
