#!/bin/bash
# HBM traffic of HoloGAN EXT-128's 5x5 input-gradient launches, the row-shared kernel (default) vs the gather loader
# (GZ_NO_DG5=1): two PMC passes each (FETCH_SIZE, WRITE_SIZE; --kernel-trace only beside --pmc).
#   bash tools/pmc_dg5.sh <tag>   -> gpurun_out/<tag>_dg5_traffic_detail.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r06}
mkdir -p gpurun_out
for variant in new old; do
  if [ $variant = old ]; then export GZ_EXPERIMENTS=1 GZ_NO_DG5=1; else unset GZ_EXPERIMENTS GZ_NO_DG5; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$variant
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_$variant -- python3 bench.py --expt hologan \
        --batch 64 --img-size 128 --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer > /dev/null 2>&1
  done
  python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE_$variant /tmp/pmc_WRITE_SIZE_$variant /tmp/dg5_traffic_$variant.json \
      gpurun_out/${tag}_dg5_traffic_detail.json hologan_ext128_bs64_$variant > /dev/null
done
unset GZ_EXPERIMENTS GZ_NO_DG5
python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_dg5_traffic_detail.json"))
for k, t in d.items():
    for lab, v in t.items():
        if "Dg" in lab:
            print(k, lab, v)
PY
