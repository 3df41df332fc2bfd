# PMC passes of the bench command (each its own run, --kernel-trace only, as MI355X_MICROARCH.md prescribes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r02b}
CMD="python bench.py --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-bs128 --no-kernel-timer"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- $CMD > /dev/null 2>&1
done
python tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE gpurun_out/${tag}_traffic.json gpurun_out/${tag}_traffic_pmc_detail.json > /dev/null
rm -rf /tmp/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- $CMD > /dev/null 2>&1
python tools/pmc_mfma_busy.py /tmp/pmc_mfma gpurun_out/${tag}_mfma_busy.json > /dev/null
rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pmc_sq -- $CMD > gpurun_out/${tag}_pmc_sq.log 2>&1
python tools/pmc_counters.py /tmp/pmc_sq gpurun_out/${tag}_sq_counters.json > /dev/null
