#!/bin/bash
# Round-5 (same passes as pmc_r04.sh; the traffic file is stamped with the source digest) PMC passes (each its own rocprofv3 run, --kernel-trace only beside --pmc): HBM traffic per launch
# (FETCH_SIZE, WRITE_SIZE; corrected as MI355X_MICROARCH.md prescribes, tools/pmc_traffic.py) and matrix-pipe busy
# (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE) for the headline configuration dc_gan bs 128 and for bs 512.
#   bash tools/pmc_r04.sh <tag>     -> gpurun_out/<tag>_traffic.json, <tag>_traffic_pmc_detail.json, <tag>_mfma_busy_bs*.json
#   OTHERS=1 adds the traffic passes of the other single-GPU configurations (wgan_gp bs 256, hologan bs 64, EXT-128,
#   wgan bs 512), so that every sub-record of the bench line carries a measured roofline.traffic
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-r05}
mkdir -p gpurun_out
cp profiles/traffic.json gpurun_out/${tag}_traffic.json 2>/dev/null
for bs in 128 512; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${c}_$bs
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_$bs -- python3 bench.py --batch $bs \
        --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer > /dev/null 2>&1
  done
  python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE_$bs /tmp/pmc_WRITE_SIZE_$bs gpurun_out/${tag}_traffic.json \
      gpurun_out/${tag}_traffic_pmc_detail.json dc_gan_bs$bs > /dev/null
  rm -rf /tmp/pmc_mfma_$bs
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma_$bs -- \
      python3 bench.py --batch $bs --steps 3 --warmup 1 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer > /dev/null 2>&1
  python3 tools/pmc_mfma_busy.py /tmp/pmc_mfma_$bs gpurun_out/${tag}_mfma_busy_bs$bs.json > /dev/null
done
if [ -n "$OTHERS" ]; then
  for spec in "wgan_gp_bs256:--expt wgan_gp --batch 256" "hologan_bs64:--expt hologan --batch 64" \
              "hologan_ext128_bs64:--expt hologan --batch 64 --img-size 128" "wgan_bs512:--expt wgan --batch 512"; do
    key=${spec%%:*}; args=${spec#*:}
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_${c}_$key
      timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${c}_$key -- python3 bench.py $args \
          --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-sub-configs --no-kernel-timer > /dev/null 2>&1
    done
    python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE_$key /tmp/pmc_WRITE_SIZE_$key gpurun_out/${tag}_traffic.json \
        gpurun_out/${tag}_traffic_pmc_detail.json $key > /dev/null
  done
fi
ls -la gpurun_out | grep ${tag}_
