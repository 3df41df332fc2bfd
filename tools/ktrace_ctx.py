"""For every dispatch of the kernels matching <substr> in a rocprofv3 kernel_trace.csv: its duration, the gap to the
previous dispatch and that dispatch's name -- to see which call sites of one kernel are slow inside a step.
    python tools/ktrace_ctx.py <kernel_trace.csv> <substr> [max rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 40
n = 0
for i, r in enumerate(rows):
    if sys.argv[2] in r["Kernel_Name"]:
        prev = rows[i - 1] if i else None
        nxt = rows[i + 1] if i + 1 < len(rows) else None
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3 if prev else 0.0
        print("%8.1f us  gap %6.1f  grid %s  after %-60s before %s" % (
            dur, gap, r.get("Grid_Size_X", r.get("Grid_Size", "?")), prev["Kernel_Name"][:60] if prev else "-",
            nxt["Kernel_Name"][:50] if nxt else "-"))
        n += 1
        if n >= lim:
            break
