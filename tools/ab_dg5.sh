#!/bin/bash
# A/B of the 5x5 input-gradient kernels inside the HoloGAN cycle (64x64 and EXT-128), two runs each way
cd $GRAFT_REPO_ROOT
for img in 128 64; do
  for sw in "" "GZ_NO_DG5=1" "" "GZ_NO_DG5=1"; do
    echo -n "img $img [$sw] "
    env GZ_EXPERIMENTS=1 $sw python3 bench.py --expt hologan --img-size $img --steps 10 --no-sub-configs --no-cpu-baseline --fid-samples 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['repetitions']['ms_per_step_each'], {k:(v['launches'],v['ms'],v['tflops']) for k,v in d['roofline']['all_igemm'].items() if 'Dg' in k and ('4x32' in k or '256x64' in k or 'Dg,256x128' in k)})"
  done
done
