# rocprofv3 --kernel-trace --stats of the bench command: the per-kernel totals / averages (kernel_stats.csv) next to the by-grid table
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02s; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json" > $O/trace.log 2>&1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$S" ]; then cp "$S" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-200; fi
ls $O/trace/* | head
rm -rf $O/trace
