#!/bin/bash
# Round-5 evidence, one gpurun call:  bash tools/profile_r05.sh <tag>
#   tools/profile_r04.sh's passes (kernel stats + gap digest per configuration) plus the data-parallel code path
#   (dc_gan bs 128 under ddp.GradSync, single-rank RCCL: --force-grad-sync) and the bench line itself.
cd $GRAFT_REPO_ROOT
tag=$1
bash tools/profile_r04.sh $tag dc_gan_bs128 dc_gan_bs512 wgan_gp_bs256 hologan_bs64 hologan_ext128_bs64 wgan_bs512 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_gs
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gs -- \
    python3 bench.py --batch 128 --force-grad-sync --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer \
    > gpurun_out/${tag}_dc_gan_bs128_gradsync_w1_prof_line.json 2> gpurun_out/${tag}_gs.err
ks=$(find /tmp/prof_gs -name "*kernel_stats.csv" | head -1); kt=$(find /tmp/prof_gs -name "*kernel_trace.csv" | head -1)
[ -n "$ks" ] && cp $ks gpurun_out/${tag}_dc_gan_bs128_gradsync_w1_kernel_stats.csv
[ -n "$kt" ] && python3 tools/gap_digest.py $kt "" 2 13 > gpurun_out/${tag}_gaps_dc_gan_bs128_gradsync_w1.txt 2>&1     # 10 + 2 + 1 pairs in the trace
python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
ls gpurun_out | grep ${tag}_ | head -40
