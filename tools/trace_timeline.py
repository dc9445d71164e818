"""Timeline of a rocprofv3 --kernel-trace CSV: start, end and duration of the order and sweep kernels of a
shuffled call in a window of the run, and the busy fraction of each kernel family (development aid).

    python tools/trace_timeline.py gpurun_out/prof_batch/batch_kernel_trace.csv [start_ms=2000] [window_ms=25]
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
start = float(sys.argv[2]) if len(sys.argv) > 2 else 2000.0
window = float(sys.argv[3]) if len(sys.argv) > 3 else 25.0
events = []
for r in rows:
    n = r["Kernel_Name"]
    if "orders" in n or "k_order_" in n:
        short = "order"
    elif "sweep_shuffled" in n:
        short = "sweep<" + n.split("<")[1].split(">")[0] + ">"
    else:
        continue
    events.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Grid_Size_X"]),
                   int(r["Workgroup_Size_X"]), r["Queue_Id"]))
events.sort()
t0 = events[0][0]
for e in events:
    if start * 1e6 <= e[0] - t0 < (start + window) * 1e6:
        print("%9.3f %9.3f %7.3f ms %-22s workgroups=%d x %d threads, queue %s" % (
            (e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2], e[3] // e[4], e[4], e[5]))
span = events[-1][1] - t0
for family in sorted({e[2] for e in events}):
    mine = sorted((e[0], e[1]) for e in events if e[2] == family)
    busy, end = 0, 0
    for a, b in mine:
        a = max(a, end)
        if b > a:
            busy += b - a
            end = b
    print("%-22s %5d launches, busy %.3f of the run (%.1f ms)" % (family, len(mine), busy / span, busy / 1e6))
