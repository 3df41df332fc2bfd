cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for args in "64 256 32 256 3 1 1" "64 128 32 256 5 2 2" "512 256 16 512 4 2 1"; do
rm -rf /tmp/wg1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wg1 -- python3 tools/wg_one.py $args > /dev/null 2>&1
echo "== $args"; python3 tools/kstats_grep.py $(find /tmp/wg1 -name "*kernel_stats.csv" | head -1) igemm reduce slab fill Memset copy
done
