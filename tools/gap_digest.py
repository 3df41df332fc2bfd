"""Idle-gap digest of one bench configuration from a rocprofv3 kernel trace (+ HIP runtime API trace).

    python tools/gap_digest.py <kernel_trace.csv> [<hip_api_trace.csv>] [batches_per_cycle]

Steady state = the second half of the trace.  For every pair of consecutive dispatches on the device the idle time
between them (next start - previous end) is attributed to the PRECEDING kernel's name.  With the API trace, a gap is
"host late" when the launch call of the next kernel returned after the previous kernel had already ended (the queue
was empty: the host is the limit) and "queued" when the launch was already in the queue (dependent-dispatch latency of
the command processor: only fewer launches remove it).  Prints per-cycle totals, the split, a histogram, and the
kernels ranked by the idle time that follows them.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\bgz::", "", name)
    name = re.sub(r"void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main():
    kt = sys.argv[1]
    ha = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else None
    rows = list(csv.DictReader(open(kt)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    api_end = {}
    if ha:
        try:
            for r in csv.DictReader(open(ha)):
                if "Launch" in r.get("Function", ""):
                    api_end[r["Correlation_Id"]] = int(r["End_Timestamp"])
        except Exception as e:  # noqa: BLE001
            print("(no usable API trace: %r)" % (e,))
    rows = rows[len(rows) // 2:]
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    # (union of the kernels' intervals: under data parallelism RCCL's kernels run on a second stream and overlap)
    busy, cur_end = 0.0, None
    for r in rows:
        a_, b_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if cur_end is None or a_ >= cur_end:
            busy += (b_ - a_) / 1e3
            cur_end = b_
        elif b_ > cur_end:
            busy += (b_ - cur_end) / 1e3
            cur_end = b_
    by = defaultdict(lambda: [0, 0.0, 0, 0.0])          # name -> [gaps, us, host-late gaps, host-late us]
    hist = defaultdict(lambda: [0, 0.0])
    edges = ((0, 1), (1, 2), (2, 3), (3, 4), (4, 6), (6, 10), (10, 20), (20, 50), (50, 1e9))
    tot = host_us = 0.0
    host_n = known = 0
    front = None                       # the kernel whose end is the latest so far (several streams may overlap)
    for a, b in zip(rows, rows[1:]):
        if front is None or int(a["End_Timestamp"]) > int(front["End_Timestamp"]):
            front = a
        a = front
        g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
        if g <= 0:
            continue
        tot += g
        e = by[short(a["Kernel_Name"])]
        e[0] += 1
        e[1] += g
        late = None
        t = api_end.get(b.get("Correlation_Id"))
        if t is not None:
            known += 1
            late = t > int(a["End_Timestamp"])
            if late:
                e[2] += 1
                e[3] += g
                host_n += 1
                host_us += g
        for lo, hi in edges:
            if lo <= g < hi:
                hist[(lo, hi)][0] += 1
                hist[(lo, hi)][1] += g
    # cycles in the window: count a kernel that runs exactly once per optimizer step
    per_cycle = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    names = defaultdict(int)
    for r in rows:
        names[r["Kernel_Name"]] += 1
    steps = sum(n for k, n in names.items() if "adam_kernel" in k or "rmsprop_kernel" in k)
    cycles = steps / per_cycle if steps else 0      # one fused optimizer launch per batch of the cycle
    if len(sys.argv) > 4:                           # the caller knows how many optimizer cycles the WHOLE trace holds
        cycles = float(sys.argv[4]) / 2.0           # (per-bucket optimizer launches under ddp.GradSync; half = this window)
    print("dispatches %d, span %.1f us, kernel time %.1f us (%.2f %% busy), idle %.1f us (%.2f %%)"
          % (len(rows), span, busy, 100 * busy / span, tot, 100 * tot / span))
    if cycles:
        print("optimizer launches %d -> ~%.1f optimizer cycles in the window: %.1f dispatches, %.3f ms kernel time, "
              "%.3f ms idle per cycle" % (steps, cycles, len(rows) / cycles, busy / cycles / 1e3, tot / cycles / 1e3))
    if known:
        print("launch already queued when the previous kernel ended: %d gaps, %.1f us (%.1f %% of idle); host late: "
              "%d gaps, %.1f us (%.1f %%)" % (known - host_n, tot - host_us, 100 * (tot - host_us) / max(tot, 1e-9),
                                              host_n, host_us, 100 * host_us / max(tot, 1e-9)))
    for (lo, hi) in edges:
        n, us = hist[(lo, hi)]
        print("  gaps %4g-%-5g us: %6d  %9.1f us" % (lo, hi, n, us))
    print("idle time by PRECEDING kernel (gaps, total us, mean us | host-late gaps, us):")
    for name, (n, us, hn, hus) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
        print("  %5d %9.1f %6.2f | %5d %8.1f  %s" % (n, us, us / n, hn, hus, name))


if __name__ == "__main__":
    main()
