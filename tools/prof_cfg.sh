# kernel trace + stats + gap digest of one bench configuration:
#   bash tools/prof_cfg.sh <tag> [bench args]      -> gpurun_out/<tag>_kernel_stats.csv, <tag>_gaps.txt, <tag>_bench.json
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/kt_$tag
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -- python3 bench.py --steps 12 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
f=$(find /tmp/kt_$tag -name "*kernel_trace.csv" | head -1)
st=$(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1)
cp $st gpurun_out/${tag}_kernel_stats.csv
python3 tools/gap_digest.py $f > gpurun_out/${tag}_gaps.txt 2>&1
