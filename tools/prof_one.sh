cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_x
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -- python bench.py --expt hologan --img-size 128 --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs > /dev/null 2>&1
python tools/kstats.py $(find /tmp/prof_x -name "*kernel_stats.csv" | head -1) 14 45 > gpurun_out/ext128_kstats.txt
