mkdir -p gpurun_out/r06d
python bench.py --no-cpu-baseline --fid-samples 0 > gpurun_out/r06d/bench_line.json 2> gpurun_out/r06d/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06d/bench_line.json"))
print(d["ms_per_step"], d["roofline"]["whole_step"]["frac"])
for k,v in d["sub_configs"].items():
    if "error" in v: print(k, v); continue
    ws=(v.get("roofline") or {}).get("whole_step") or {}
    ge=v.get("grad_exchange") or {}
    print(k, v.get("ms_per_step"), ws.get("frac"), v.get("vs_plain"), ge.get("buckets"), (ge.get("overlap") or {}).get("exposed_wait_ms_per_step"))
PY
