#!/bin/bash
# A/B of one bench configuration under experiment switches, three repetitions each:
#   gpurun -- bash tools/ab_expt.sh <tag> <expt> <batch> <steps> "<SW1=..>" "<SW2=.. SW3=..>" ...
cd $GRAFT_REPO_ROOT
tag=$1; expt=$2; bs=$3; steps=$4; shift 4
out=gpurun_out/${tag}_ab_${expt}_bs${bs}.txt
: > $out
for sw in "" "$@" ""; do
  echo "== [$sw]" >> $out
  env GZ_EXPERIMENTS=1 $sw python3 bench.py --expt $expt --batch $bs --steps $steps --no-sub-configs --no-cpu-baseline --no-gradsync-w1 --fid-samples 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repetitions']['ms_per_step_each'])" >> $out
done
cat $out
