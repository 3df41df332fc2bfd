# A/B of an environment toggle inside the bench step: per-kernel averages from rocprofv3
#   bash tools/pf_ab.sh VAR val1 val2 ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
var=$1; shift
export GZ_EXPERIMENTS=1      # the GZ_* switches are only read then (csrc/gz_knobs.h)
for v in "$@"; do
  export $var=$v
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$v -- python3 bench.py --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs > gpurun_out/pf_$v.json 2>/dev/null
  f=$(find /tmp/pf_$v -name "*kernel_stats.csv" | head -1)
  echo "== $var=$v $(python3 -c "import json;print(json.loads(open('gpurun_out/pf_$v.json').read().strip().splitlines()[-1])['ms_per_step'])")"
  python3 tools/kstats_grep.py $f smallc4 "ConvFwdALoaderRow4<128>" "WgALoaderRow<64>" "WgALoaderRow<128>, gz::WgBLoaderRow<64" edge
done
