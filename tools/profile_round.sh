cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
for e in dc_gan gan_stability_r1 hologan wgan_gp; do
  rm -rf /tmp/prof_$e
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$e -- python bench.py --expt $e --steps 10 --warmup 2 --no-cpu-baseline --no-bs128 > gpurun_out/${e}_prof_line.json 2>/dev/null
  cp $(find /tmp/prof_$e -name "*kernel_stats.csv" | head -1) gpurun_out/r01i_${e}_kernel_stats.csv
  python tools/kstats.py gpurun_out/r01i_${e}_kernel_stats.csv 13 3
done
bash tools/bench_all.sh
cat gpurun_out/bench_default.json | cut -c1-1200
