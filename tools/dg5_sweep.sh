cd $GRAFT_REPO_ROOT
for w in 512 256 384 768 1024; do for mc in 24 48; do
  echo "== GZ_DG5_WGS=$w GZ_DG5_MIN_CHUNKS=$mc"
  GZ_EXPERIMENTS=1 GZ_DG5_WGS=$w GZ_DG5_MIN_CHUNKS=$mc timeout 300 python tools/tap_bench.py 64 2>&1 | grep "holo" | grep -v 1x1 | sed -e 's/ | Wg.*//' -e 's/.*GF  F [0-9x]* *[0-9.]* ms *[0-9.]* TF | //'
done; done
echo "== bs 128"; timeout 300 python tools/tap_bench.py 128 2>&1 | grep "holo128" | sed -e 's/ | Wg.*//'
