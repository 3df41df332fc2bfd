mkdir -p gpurun_out/r06c
python -m pytest tests/test_ddp_gpu.py tests/test_grad_sinks_gpu.py -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k hologan 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/kt_f
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_f -- python3 bench.py --steps 8 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer --force-grad-sync > gpurun_out/r06c/dc_sync.json 2> gpurun_out/r06c/dc_sync.err
f=$(find /tmp/kt_f -name "*kernel_trace.csv" | head -1)
python3 tools/fill_trace.py $f > gpurun_out/r06c/fill_trace.txt 2>&1
head -60 gpurun_out/r06c/fill_trace.txt
python bench.py --no-cpu-baseline --fid-samples 0 > gpurun_out/r06c/bench_line.json 2> gpurun_out/r06c/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r06c/bench_line.json"))
print(d["ms_per_step"], d["roofline"]["whole_step"]["frac"])
for k,v in d["sub_configs"].items():
    if "error" in v: print(k, v); continue
    ws=(v.get("roofline") or {}).get("whole_step") or {}
    ge=v.get("grad_exchange") or {}
    print(k, v.get("ms_per_step"), ws.get("frac"), v.get("vs_plain"), ge.get("buckets"), (ge.get("overlap") or {}).get("exposed_wait_ms_per_step"))
PY
