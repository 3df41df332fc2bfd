"""Where do the __amd_rocclr_fillBufferAligned / copyBuffer dispatches of a GradSync run come from?  Reads a rocprofv3
kernel trace CSV: per queue, the dispatch counts; for every fill / copy the kernel that ran right before it on the SAME
queue and the nearest RCCL / gz kernel in time.     python tools/fill_trace.py <kernel_trace.csv>"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
print("columns:", list(rows[0].keys()))
qcol = "Queue_Id" if "Queue_Id" in rows[0] else None
scol = "Stream_Id" if "Stream_Id" in rows[0] else None


def short(n):
    return n.replace("void ", "").replace("gz::", "")[:70]


per_q = Counter((r.get(qcol), r.get(scol)) for r in rows)
print("dispatches per (queue, stream):", per_q.most_common())
last_on_q = {}
before = Counter()
sizes = Counter()
for r in rows:
    q = (r.get(qcol), r.get(scol))
    name = r["Kernel_Name"]
    if "rocclr" in name:
        prev = last_on_q.get(q)
        before[(short(name), q, short(prev) if prev else None)] += 1
        sizes[(short(name), r.get("Grid_Size"), r.get("Workgroup_Size"))] += 1
    last_on_q[q] = name
for k, c in before.most_common(30):
    print("%4d  %s" % (c, k))
print("grid sizes:")
for k, c in sizes.most_common(20):
    print("%4d  %s" % (c, k))
