"""Print calls / average duration of the kernels whose name matches any of the given substrings, from a rocprofv3
kernel_stats.csv:   python tools/kstats_grep.py <csv> <substr> [<substr> ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    name = r.get("Name") or r.get("KernelName") or ""
    if any(s in name for s in sys.argv[2:]):
        print("%8s calls %9.1f us avg  %s" % (r.get("Calls"), float(r.get("AverageNs", 0)) / 1e3, name[:150]))
