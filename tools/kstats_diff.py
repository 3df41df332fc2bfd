"""Side-by-side per-cycle kernel time of two rocprofv3 kernel_stats.csv files:  python tools/kstats_diff.py A.csv B.csv cyclesA [cyclesB]"""
import csv, re, sys
def load(p, cycles):
    out = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*$", "", r["Name"].replace("gz::", "").replace("void ", ""))[:100]
        c, t = out.get(n, (0, 0.0))
        out[n] = (c + int(r["Calls"]) / cycles, t + float(r["TotalDurationNs"]) / cycles / 1e3)
    return out
ca = float(sys.argv[3]); cb = float(sys.argv[4]) if len(sys.argv) > 4 else ca
a, b = load(sys.argv[1], ca), load(sys.argv[2], cb)
keys = sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, (0, 0))[1] - b.get(k, (0, 0))[1]))
ta = sum(v[1] for v in a.values()); tb = sum(v[1] for v in b.values())
print("total us/cycle: A %.1f  B %.1f  (A-B %.1f); launches/cycle A %.1f B %.1f" % (ta, tb, ta - tb, sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
for k in keys[:int(sys.argv[5]) if len(sys.argv) > 5 else 30]:
    va, vb = a.get(k, (0, 0)), b.get(k, (0, 0))
    print("%7.1f us (%5.2f x) | %7.1f us (%5.2f x) | d %+7.1f | %s" % (va[1], va[0], vb[1], vb[0], va[1] - vb[1], k))
