cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for e in gan_stability_r1 hologan wgan_gp; do
  timeout 300 python bench.py --expt $e --steps 10 --warmup 3 > gpurun_out/${e}_bench.json 2> gpurun_out/${e}_bench.err
  rm -rf /tmp/prof_$e
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$e -- python bench.py --expt $e --steps 10 --warmup 3 --no-kernel-timer > /dev/null 2>&1
  cp $(find /tmp/prof_$e -name "*kernel_stats.csv" | head -1) gpurun_out/${e}_kernel_stats.csv
  python - <<PY
import json
d=json.load(open("gpurun_out/${e}_bench.json"))
print("$e", d["value"], d["ms_per_step"], d["roofline"]["whole_step"])
PY
done
