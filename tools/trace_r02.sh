cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02b
rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json" > $O/trace.log 2>&1)
T=$(find $O/trace -name "*.db" | head -1)
python tools/prof_summary.py $T 60 --by-grid > $O/kernels_by_grid.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/trace
head -64 $O/kernels_by_grid.txt
