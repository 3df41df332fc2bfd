#!/bin/bash
# A/B of the metric configuration (dc_gan bs 128) under experiment switches, three repetitions each:
#   gpurun -- bash tools/ab_bs128.sh <tag> "<SW1=..>" "<SW2=.. SW3=..>" ...
cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/${tag}_ab_bs128.txt
: > $out
for sw in "" "$@"; do
  echo "== [$sw]" >> $out
  env GZ_EXPERIMENTS=1 $sw python3 bench.py --batch 128 --no-sub-configs --no-cpu-baseline --no-gradsync-w1 --fid-samples 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repetitions']['ms_per_step_each'])" >> $out
done
cat $out
