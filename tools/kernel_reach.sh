#!/bin/bash
# Which kernel symbols do the GPU test suite and the bench configurations actually launch?  (the
# library holds ~800 kernel instantiations.)   gpurun -- bash tools/kernel_reach.sh <tag>
#   -> gpurun_out/<tag>_reached_kernels.txt (demangled names, one per line, with the number of launches)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=$1
rm -rf /tmp/reach; mkdir -p /tmp/reach
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/reach/tests -- python3 -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/${tag}_reach_pytest.txt 2>&1
i=0
for a in "--expt dc_gan --batch 128" "--expt dc_gan --batch 512" "--expt dc_gan --batch 64" "--expt wgan_gp --batch 256" "--expt wgan --batch 512" \
         "--expt hologan --batch 64" "--expt hologan --batch 64 --img-size 128" "--expt gan_stability_r1 --batch 64" "--expt dc_gan --batch 128 --force-grad-sync"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/reach/b$i -- python3 bench.py $a --steps 2 --warmup 1 --reps 1 \
      --no-cpu-baseline --no-sub-configs --no-kernel-timer > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/reach/fid -- python3 bench.py --batch 128 --steps 1 --warmup 1 --reps 1 \
    --no-cpu-baseline --no-gradsync-w1 --fid-samples 500 > /dev/null 2>&1
python3 - <<PY > gpurun_out/${tag}_reached_kernels.txt
import csv, glob, collections
c = collections.Counter()
for f in glob.glob('/tmp/reach/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        c[r['Name']] += int(r['Calls'])
for k, v in sorted(c.items()):
    print("%8d  %s" % (v, k))
PY
wc -l gpurun_out/${tag}_reached_kernels.txt; tail -3 gpurun_out/${tag}_reach_pytest.txt
