#!/bin/bash
# HoloGAN's rigid resampling with / without the LDS bank-group swizzle (GZ_NO_RESAMPLE_SWIZZLE=1): the resample tests,
# then in-step durations of the two staged kernels and the cycle time, both ways, in one call.
#   gpurun -- bash tools/resample_ab.sh <tag>
cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/${tag}_resample_ab.txt
: > $out
timeout 600 python3 -m pytest tests -x -q -m gpu -k "resample or index_golden or hologan_step or hologan_bs64" 2>&1 | tail -2 >> $out
for sw in "GZ_NO_RESAMPLE_SWIZZLE=1" ""; do
  echo "== [$sw]" >> $out
  unset GZ_NO_RESAMPLE_SWIZZLE; export GZ_EXPERIMENTS=1 $sw
  bash tools/prof_cfg.sh ${tag}_x --expt hologan --batch 64
  python3 tools/kstats_grep.py gpurun_out/${tag}_x_kernel_stats.csv resample | cut -c1-110 >> $out
  python3 bench.py --expt hologan --batch 64 --steps 10 --no-sub-configs --no-cpu-baseline --no-gradsync-w1 --fid-samples 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repetitions']['ms_per_step_each'])" >> $out
done
cat $out
