cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02a}
for e in hologan gan_stability_r1 wgan_gp; do
  rm -rf /tmp/prof_$e
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$e -- python bench.py --expt $e --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-bs128 > gpurun_out/${tag}_${e}_prof_line.json 2>/dev/null
  cp $(find /tmp/prof_$e -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_${e}_kernel_stats.csv
  python tools/kstats.py gpurun_out/${tag}_${e}_kernel_stats.csv 14 45 > gpurun_out/${tag}_${e}_kstats.txt
done
