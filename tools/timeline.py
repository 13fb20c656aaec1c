"""Timeline of the LAST step in a rocprofv3 --kernel-trace csv: every kernel in start order with its duration, the idle gap before it and its queue.
    python tools/timeline.py <kernel_trace.csv> <out.txt> [first-kernel-substring, default march_kernel] [kernels to keep before it, default 8]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[3] if len(sys.argv) > 3 else "march_kernel"
back = int(sys.argv[4]) if len(sys.argv) > 4 else 8
i0 = max(0, max(i for i, r in enumerate(rows) if mark in r["Kernel_Name"]) - back)
t0 = int(rows[i0]["Start_Timestamp"])
queues = {}
end_all, tot, busy_until, idle = t0, 0, t0, 0
with open(sys.argv[2], "w") as o:
    for r in rows[i0:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = queues.setdefault(r.get("Queue_Id", "0"), len(queues))
        gap = s - busy_until                      # idle time of the WHOLE device before this kernel (no queue busy)
        if gap > 0:
            idle += gap
        busy_until = max(busy_until, e)
        tot += e - s
        o.write("%9.1f us  dur %8.1f  idle-before %6.1f  q%d  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, max(gap, 0) / 1e3, q, r["Kernel_Name"].split("(")[0][:80]))
    o.write("kernels %.1f us (sum over queues), device idle %.1f us, span %.1f us, launches %d, queues %d\n" % (tot / 1e3, idle / 1e3, (busy_until - t0) / 1e3, len(rows) - i0, len(queues)))
print(open(sys.argv[2]).read()[-7000:])
