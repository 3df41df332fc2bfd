#!/bin/bash
# In-step durations of the edge-layer kernels (inputs cold, as inside the G+D cycle) with the direct kernels on / off:
#   gpurun -- bash tools/edge_instep.sh <tag> [batch]
cd $GRAFT_REPO_ROOT
tag=$1; bs=${2:-128}
out=gpurun_out/${tag}_edge_instep_bs${bs}.txt
: > $out
for sw in "GZ_NO_FEWC_WG=1" "GZ_FEWC_WG_BLOCKS=512" "GZ_FEWC_WG_BLOCKS=256"; do
  echo "== $sw" >> $out
  unset GZ_NO_FEWC_WG GZ_FEWC_WG_BLOCKS; export GZ_EXPERIMENTS=1 $sw
  bash tools/prof_cfg.sh ${tag}_x --batch $bs
  python3 -c "import json;d=json.loads(open('gpurun_out/${tag}_x_bench.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])" >> $out
  python3 tools/kstats_grep.py gpurun_out/${tag}_x_kernel_stats.csv fewc smallc4 "WgALoaderRow<" "ConvFwdALoaderRow4" reduce_multi adam_from_slabs >> $out
done
cat $out
