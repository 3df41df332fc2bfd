cd $GRAFT_REPO_ROOT
for t in -1 0 1 3; do echo -n "GZ_TILE=$t "; GZ_EXPERIMENTS=1 GZ_TILE=$t python3 tools/fid_prof.py 5000 250 2>&1 | tail -1; done
for b in 500 1000; do echo -n "batch $b "; python3 tools/fid_prof.py 10000 $b 2>&1 | tail -2; done
