"""Idle time between consecutive dispatches in a rocprofv3 kernel_trace.csv: total, histogram, and the largest gaps with
their neighbours (second half of the trace = steady state):   python tools/ktrace_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
gaps = []
for a, b in zip(rows, rows[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    gaps.append((g, a["Kernel_Name"][:70], b["Kernel_Name"][:70]))
tot = sum(max(g[0], 0) for g in gaps)
print("dispatches %d, span %.1f us, kernel time %.1f us (%.1f %%), gaps %.1f us" % (len(rows), span, busy, 100 * busy / span, tot))
for lo, hi in ((0, 1), (1, 3), (3, 6), (6, 15), (15, 50), (50, 1e9)):
    sel = [g[0] for g in gaps if lo <= g[0] < hi]
    print("  gaps %4g-%-4g us: %5d, %.1f us" % (lo, hi, len(sel), sum(sel)))
for g in sorted(gaps, reverse=True)[:12]:
    print("  %7.1f us  after %-70s before %s" % g)
