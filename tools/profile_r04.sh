#!/bin/bash
# Round-4 evidence, one gpurun call:  bash tools/profile_r04.sh <tag> [configs...]
#   per bench configuration: rocprofv3 --kernel-trace --stats of the bench command
#   -> profiles-ready files under gpurun_out/<tag>_*: kernel stats (csv) and the idle-gap digest (tools/gap_digest.py:
#   idle time by preceding kernel + histogram), followed by the host's enqueue time per cycle measured WITHOUT the
#   profiler (tools/host_profile.py --brief): the HIP API trace costs the host more per launch than the launch itself
#   (r04a: 32 % idle under --hip-runtime-trace against 5-7 % without), so it cannot say whether the host was late.
#   HIPTRACE=1 adds that pass anyway (correlation-id join: "queued" vs "host late" per gap).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
cfgs=${@:-"dc_gan_bs128 dc_gan_bs512 wgan_gp_bs256 hologan_bs64 hologan_ext128_bs64"}
mkdir -p gpurun_out
for c in $cfgs; do
  case $c in
    dc_gan_bs128) a="--expt dc_gan --batch 128"; cyc=2;;
    dc_gan_bs512) a="--expt dc_gan --batch 512"; cyc=2;;
    wgan_gp_bs256) a="--expt wgan_gp --batch 256"; cyc=2;;
    wgan_bs512) a="--expt wgan --batch 512"; cyc=6;;
    hologan_bs64) a="--expt hologan --batch 64"; cyc=3;;
    hologan_ext128_bs64) a="--expt hologan --batch 64 --img-size 128"; cyc=3;;
    *) echo "unknown config $c"; continue;;
  esac
  rm -rf /tmp/prof_$c
  case $c in dc_gan*|wgan*) e=${c%%_bs*};; *) e=hologan;; esac
  timeout 400 rocprofv3 --kernel-trace ${HIPTRACE:+--hip-runtime-trace} --stats --output-format csv -d /tmp/prof_$c -- \
      python3 bench.py $a --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer \
      > gpurun_out/${tag}_${c}_prof_line.json 2> gpurun_out/${tag}_${c}_prof.err
  ks=$(find /tmp/prof_$c -name "*kernel_stats.csv" | head -1)
  kt=$(find /tmp/prof_$c -name "*kernel_trace.csv" | head -1)
  ha=$(find /tmp/prof_$c -name "*hip_api_trace.csv" | head -1)
  [ -n "$ks" ] && cp $ks gpurun_out/${tag}_${c}_kernel_stats.csv
  [ -n "$kt" ] && python3 tools/gap_digest.py $kt "$ha" $cyc > gpurun_out/${tag}_gaps_${c}.txt 2>&1
  python3 tools/host_profile.py $(echo $a | sed -e 's/--expt //' -e 's/--batch //' -e 's/--img-size //') --brief \
      >> gpurun_out/${tag}_gaps_${c}.txt 2>/dev/null
done
ls -la gpurun_out | tail -30
