# round-2 evidence: rocprofv3 kernel stats of the bench command at bs 512 and bs 128 (-> profiles/r02_*), then the bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02a}
for b in 512 128; do
  rm -rf /tmp/prof_$b
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$b -- python bench.py --batch $b --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-bs128 > gpurun_out/${tag}_bs${b}_prof_line.json 2>/dev/null
  cp $(find /tmp/prof_$b -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_bs${b}_kernel_stats.csv
  python tools/kstats.py gpurun_out/${tag}_bench_bs${b}_kernel_stats.csv 14 40 > gpurun_out/${tag}_bs${b}_kstats.txt
done
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.err
