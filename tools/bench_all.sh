# bench lines of every experiment (no profiler): bash tools/bench_all.sh [extra bench.py flags]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for e in dc_gan wgan wgan_gp hologan gan_stability_r1; do
  timeout 400 python bench.py --expt $e --steps 10 --warmup 3 --no-kernel-timer --no-cpu-baseline "$@" > gpurun_out/${e}_line.json 2> gpurun_out/${e}_line.err
  python - <<PY
import json
d=json.load(open("gpurun_out/${e}_line.json"))
print("$e", d["value"], "img/s", d["ms_per_step"], "ms", (d.get("sub_configs") or {}).get("dc_gan_bs512", {}).get("ms_per_step"))
PY
done
