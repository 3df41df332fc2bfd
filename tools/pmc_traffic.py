"""HBM bytes per launch of every implicit-GEMM kernel from two rocprofv3 PMC passes.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirA> -- python bench.py --batch 128 --steps 3 \
        --warmup 1 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer
    rocprofv3 --pmc WRITE_SIZE ... -d <dirB> -- (same)
    python tools/pmc_traffic.py <dirA> <dirB> profiles/traffic.json profiles/<round>_traffic_pmc_detail.json dc_gan_bs128

The last argument is bench.py's configuration key (SUB_CONFIGS): traffic.json holds one table per configuration.

traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE under-reports
reads by 2x on gfx950; the raw sum is kept in the detail file).  Labels match bench.py's roofline labels.
"""
import csv
import glob
import json
import os
import re
import sys


def label_of(name):
    m = re.search(r"igemm_kernel<gz::TileCfg<(\d+), (\d+), (\d+), (\d+)>, gz::(\w+)<", name)
    if not m:
        # round 3: igemm2_kernel / igemm2r_kernel<gz::TileCfg2<WM, WN, TN, OCC>, loader...> (TM = 4)
        # (round 6: TileCfg2 has a fifth argument, TM, printed as ", 4>" / ", 2>")
        m2 = re.search(r"igemm2r?_kernel<gz::TileCfg2<(\d+), (\d+), (\d+), (\d+)(?:, \d+)?>, gz::(\w+)<", name)
        mw = re.search(r"igemm2w_kernel<gz::TileCfg2<(\d+), (\d+), (\d+), (\d+)(?:, \d+)?>, ", name)
        if mw:        # weight gradient with both operands by LDS-DMA: no loader types in the name
            return "igemm<Wg,%dx%d>" % (int(mw.group(1)) * 128, int(mw.group(2)) * int(mw.group(3)) * 32)
        if not m2:
            return None
        wm, wn, tn = (int(m2.group(i)) for i in range(1, 4))
        tm, loader = 4, m2.group(5)
    else:
        wm, wn, tm, tn = (int(m.group(i)) for i in range(1, 5))
        loader = m.group(5)
    if loader.startswith("ConvFwd") or loader.startswith("ConvTap"):
        op = "F"
    elif loader.startswith("ConvDg"):
        op = "Dg"
    elif loader.startswith("WgALoader"):
        op = "Wg"
    else:
        return None
    if loader == "ConvDg5A2":          # (round 6) 256 pixels x (2 column phases x 64 channels): bench.py's label
        return "igemm<Dg,256x(4x32)>"
    return "igemm<%s,%dx%d>" % (op, wm * tm * 32, wn * tn * 32)


def collect(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            lab = label_of(row["Kernel_Name"])
            if lab:
                n, tot = out.get(lab, (0, 0.0))
                out[lab] = (n + 1, tot + float(row["Counter_Value"]))
    return out


if __name__ == "__main__":
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    traffic, detail = {}, {}
    for lab in sorted(set(fetch) & set(write)):
        nf, f = fetch[lab]
        nw, w = write[lab]
        fk, wk = f / nf, w / nw
        traffic[lab] = int((2 * fk + wk) * 1024)
        detail[lab] = {"launches_sampled": nf, "fetch_kb_raw": round(fk, 1), "write_kb": round(wk, 1),
                       "hbm_bytes_per_launch_corrected": traffic[lab], "hbm_bytes_per_launch_raw": int((fk + wk) * 1024)}
    key = sys.argv[5] if len(sys.argv) > 5 else "dc_gan_bs512"
    try:
        table = json.load(open(sys.argv[3]))
    except Exception:  # noqa: BLE001
        table = {}
    table[key] = traffic
    # stamp the table with the kernel sources it was measured on (bench.py: roofline.traffic_stale when the running
    # library was built from other sources)
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from lightning_gan_zoo_amd.build import source_digest
        stamps = table.get("_source_digest")
        if not isinstance(stamps, dict):
            stamps = {}
        stamps[key] = source_digest()
        table["_source_digest"] = stamps
    except Exception as e:  # noqa: BLE001
        print("(no source digest: %r)" % (e,), file=sys.stderr)
    json.dump(table, open(sys.argv[3], "w"), indent=1)
    try:
        alld = json.load(open(sys.argv[4]))
    except Exception:  # noqa: BLE001
        alld = {}
    alld[key] = detail
    json.dump(alld, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(detail, indent=1))
