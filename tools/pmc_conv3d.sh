cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/pm1 /tmp/pm2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pm1 -- python tools/conv3d_bench.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pm2 -- python tools/conv3d_bench.py > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
for d in ("/tmp/pm1","/tmp/pm2"):
    agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for p in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"]
            if "igemm" not in k: continue
            short=k[k.find("gz::Conv3D") if "Conv3D" in k else k.find("gz::Wg"):][:60]
            a=agg[short][r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
    for k,v in agg.items():
        print(k, {c:(round(x[1]/x[0]/1e6,2)) for c,x in v.items()}, "(M per launch)")
PY
