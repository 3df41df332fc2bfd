# kernel trace of the bench step; context of one kernel's dispatches or the idle gaps:
#   bash tools/ktrace_run.sh <substr>|--gaps [bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
pat=$1; shift
rm -rf /tmp/kt
timeout 250 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 bench.py --steps 6 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs "$@" > /dev/null 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
if [ "$pat" = "--gaps" ]; then python3 tools/ktrace_gaps.py $f; else python3 tools/ktrace_ctx.py $f "$pat" ${KT_ROWS:-30}; fi
