# round-3 evidence: rocprofv3 kernel stats of the bench command (bs 512, bs 128, wgan_gp, hologan, EXT-128), the PMC
# passes (traffic, MFMA busy, SQ counters: each its own run, --kernel-trace only) and the bench line of the same code
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r03}
mkdir -p gpurun_out/$tag
prof() {   # name, bench args...
  name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python bench.py "$@" --steps 10 --warmup 2 --reps 1 --no-cpu-baseline --no-sub-configs > gpurun_out/$tag/${name}_prof_line.json 2>/dev/null
  cp $(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/${tag}_${name}_kernel_stats.csv
  python tools/kstats.py gpurun_out/$tag/${tag}_${name}_kernel_stats.csv 14 40 > gpurun_out/$tag/${name}_kstats.txt
}
prof bench_bs512 --batch 512
prof bench_bs128 --batch 128
prof wgan_gp --expt wgan_gp
prof hologan --expt hologan
prof hologan_ext128 --expt hologan --img-size 128
CMD="python bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- $CMD > /dev/null 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/$tag/${tag}_traffic.json gpurun_out/$tag/${tag}_traffic_pmc_detail.json > /dev/null
rm -rf /tmp/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- $CMD > /dev/null 2>&1
python tools/pmc_mfma_busy.py /tmp/pmc_mfma gpurun_out/$tag/${tag}_mfma_busy.json > /dev/null
rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_sq -- $CMD > gpurun_out/$tag/pmc_sq.log 2>&1
python tools/pmc_counters.py /tmp/pmc_sq gpurun_out/$tag/${tag}_sq_counters.json > /dev/null
python bench.py > gpurun_out/$tag/${tag}_bench_line.json 2> gpurun_out/$tag/bench.err
