cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/fidp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fidp -- python3 tools/fid_prof.py 5000 250 > gpurun_out/r06_fid_prof.json 2>/dev/null
cp $(find /tmp/fidp -name "*kernel_stats.csv" | head -1) gpurun_out/r06_fid_kernel_stats.csv
cat gpurun_out/r06_fid_prof.json
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open("gpurun_out/r06_fid_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    n=re.sub(r"\bgz::","",r["Name"]); n=re.sub(r"^void ","",n); n=re.sub(r"\(.*$","",n)
    print("%5.1f%% calls %6s avg %8.1f us  %s"%(100*float(r["TotalDurationNs"])/tot, r["Calls"], float(r["AverageNs"])/1e3, n[:130]))
PY
for bsz in 250 500 1000; do python3 tools/fid_prof.py 10000 $bsz 2>/dev/null | tail -1; done
