#!/bin/bash
# The 3-channel edge layers (tools/edge_bench.py) across batch sizes, with the direct kernels' switches:
#   dgrad_smallc4 (64|128 -> 3 transposed convolution): wavefronts per 64 lane positions (GZ_SMALLC_KS)
#   wgrad_k4s2p1_fewc (64|128 x 48 weight gradient): on / off (GZ_NO_FEWC_WG), slabs (GZ_FEWC_WG_BLOCKS)
#   gpurun -- bash tools/smallc_sweep.sh <tag> [quick]
cd $GRAFT_REPO_ROOT
tag=${1:-smallc}
out=gpurun_out/${tag}_smallc_sweep.txt
: > $out
timeout 900 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv or smallc or edge or transpose or few" 2>&1 | tail -5 >> $out
for bs in 64 128 256 512; do
  if [ -z "$2" ]; then
    for ks in 0 1 4 8; do
      echo "== bs $bs KS $ks" >> $out
      GZ_EXPERIMENTS=1 GZ_SMALLC_KS=$ks timeout 120 python3 tools/edge_bench.py $bs 2>&1 | grep "^Dg\|^bs" >> $out
    done
  fi
  for sw in "GZ_NO_FEWC_WG=1" "GZ_FEWC_WG_BLOCKS=256" "GZ_FEWC_WG_BLOCKS=512" "GZ_FEWC_WG_BLOCKS=1024"; do
    echo "== bs $bs $sw" >> $out
    env GZ_EXPERIMENTS=1 $sw timeout 120 python3 tools/edge_bench.py $bs 2>&1 | grep "^F \|^Wg" >> $out
  done
done
cat $out
