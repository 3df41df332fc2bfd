"""MFMA-busy fraction per implicit-GEMM kernel from one rocprofv3 PMC pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d <dir> -- \
        python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-bs128 --no-kernel-timer
    python tools/pmc_mfma_busy.py <dir> profiles/<round>_mfma_busy.json

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs
(256 CU x 4); GRBM_GUI_ACTIVE is the dispatch's active cycles summed over the 8 XCDs (MI355X_MICROARCH.md).
busy fraction = MFMA_BUSY / (1024 x GRBM_GUI_ACTIVE / 8).
"""
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
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
        continue
    n, busy = d["SQ_VALU_MFMA_BUSY_CYCLES"]
    _, act = d["GRBM_GUI_ACTIVE"]
    out[lab] = {"launches_sampled": n, "mfma_busy_cycles_per_launch": round(busy / n), "gui_active_per_launch": round(act / n),
                "mfma_busy_fraction": round(busy / (1024.0 * act / 8.0), 4)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
