cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06sb; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr -o t -- bash -c "cd $GRAFT_REPO_ROOT && python tools/stress_bench.py > $O/stress.txt" > $O/tr.log 2>&1)
T=$(find $O/tr -name "*.db" | head -1)
python tools/prof_summary.py $T 30 --by-grid > $O/by_grid.txt
rm -rf $O/tr
