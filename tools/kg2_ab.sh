# A/B of the two-wave-group workgroups (igemm2 KG = 2): per-layer rates at bs 128 / 256 with and without, stats epilogues
cd $GRAFT_REPO_ROOT
for bs in 128 256; do
  for v in 0 1; do
    if [ $v = 1 ]; then export GZ_EXPERIMENTS=1 GZ_NO_KG2=1; else unset GZ_EXPERIMENTS GZ_NO_KG2; fi
    echo "== bs $bs no_kg2=$v"
    python tools/conv_bench2.py $bs f,d --stats 2>&1 | grep -v amdgpu.ids
  done
done
