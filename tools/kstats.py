"""Summarise a rocprofv3 *_kernel_stats.csv per optimizer cycle: python tools/kstats.py file.csv [cycles] [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
cycles = float(sys.argv[2]) if len(sys.argv) > 2 else 14
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("kernel time per cycle: %.3f ms (%d kernels names)" % (tot / 1e6 / cycles, len(rows)))
for r in rows[:top]:
    n = re.sub(r"^void ", "", r["Name"])
    n = re.sub(r"\(.*", "", n).replace("gz::", "")[:130]
    print("%8.3f ms %6.1f calls %8.1f us  %s" % (int(r["TotalDurationNs"]) / 1e6 / cycles, int(r["Calls"]) / cycles,
                                               float(r["AverageNs"]) / 1e3, n))
