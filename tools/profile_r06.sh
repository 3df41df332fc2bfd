#!/bin/bash
# Round-6 evidence, one gpurun call:  bash tools/profile_r06.sh <tag>
#   profile_r05.sh's passes plus config 5 under the data-parallel code path (hologan EXT-128, single-rank RCCL) and the
#   PMC passes (HBM traffic per launch for every configuration, matrix-pipe busy for the two dc_gan batches).
cd $GRAFT_REPO_ROOT
tag=${1:-r06}
bash tools/profile_r04.sh $tag dc_gan_bs128 dc_gan_bs512 wgan_gp_bs256 hologan_bs64 hologan_ext128_bs64 wgan_bs512 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for spec in "dc_gan_bs128:--batch 128:2" "hologan_ext128_bs64:--expt hologan --batch 64 --img-size 128:3"; do
  key=${spec%%:*}; rest=${spec#*:}; args=${rest%%:*}; cyc=${rest##*:}
  rm -rf /tmp/prof_gs
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gs -- \
      python3 bench.py $args --force-grad-sync --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer \
      > gpurun_out/${tag}_${key}_gradsync_w1_prof_line.json 2> gpurun_out/${tag}_gs.err
  ks=$(find /tmp/prof_gs -name "*kernel_stats.csv" | head -1); kt=$(find /tmp/prof_gs -name "*kernel_trace.csv" | head -1)
  [ -n "$ks" ] && cp $ks gpurun_out/${tag}_${key}_gradsync_w1_kernel_stats.csv
  [ -n "$kt" ] && python3 tools/gap_digest.py $kt "" $cyc 13 > gpurun_out/${tag}_gaps_${key}_gradsync_w1.txt 2>&1
done
OTHERS=1 bash tools/pmc_r05.sh $tag > /dev/null 2>&1
# the bench line reads profiles/traffic.json: hand it this call's PMC table (stamped with the sources' digest) first
cp gpurun_out/${tag}_traffic.json profiles/traffic.json
python3 bench.py --detail > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
ls gpurun_out | grep "^${tag}_" | head -60
