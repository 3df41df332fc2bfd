cd $GRAFT_REPO_ROOT
for cfg in "512 1024" "1024 1024" "1024 2048" "2048 2048" "256 1024"; do
  set -- $cfg
  export GZ_EXPERIMENTS=1 GZ_SPLIT_BELOW=$1 GZ_SPLIT_TARGET=$2
  for e in "hologan" "gan_stability_r1" "wgan_gp" "dc_gan --batch 128" "dc_gan"; do
    python bench.py --expt $e --steps 10 --warmup 3 --no-kernel-timer --no-cpu-baseline --no-bs128 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('below $1 target $2', '$e', d['ms_per_step'])"
  done
done
