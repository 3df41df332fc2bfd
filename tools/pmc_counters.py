"""Per-label averages of arbitrary rocprofv3 PMC counters over the implicit-GEMM launches of one run:

    python tools/pmc_counters.py <dir> <out.json>

Labels as in bench.py's roofline ("igemm<Dg,128x128>" ...).  Derived ratios for the SQ set of tools/pmc_round2.sh:
SQ_* wave counters are in quad-cycles summed over waves (MI355X_MICROARCH.md); WAIT_ANY + WAIT_INST_ANY +
ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import label_of  # noqa: E402

agg = {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        lab = label_of(row["Kernel_Name"])
        if not lab:
            continue
        d = agg.setdefault(lab, {})
        n, tot = d.get(row["Counter_Name"], (0, 0.0))
        d[row["Counter_Name"]] = (n + 1, tot + float(row["Counter_Value"]))
out = {}
for lab, d in sorted(agg.items()):
    o = {"launches_sampled": max(n for n, _ in d.values())}
    for k, (n, tot) in d.items():
        o[k] = round(tot / n, 1)
    wc = o.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if k in o:
                o[k + "/WAVE_CYCLES"] = round(o[k] / wc, 4)
    out[lab] = o
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
