cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
(cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $O/ttrace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --mode train-graph --steps 30 --warmup 5 > $O/bench_train_graph_under_rocprof.json" > $O/ttrace.log 2>&1)
TT=$(find $O/ttrace -name "*.db" | head -1)
python tools/prof_summary.py $TT --by-family --window-ms 250 0 > $O/train_graph_kernels_by_family.txt
python tools/prof_summary.py $TT 80 --by-grid --window-ms 250 0 > $O/train_graph_kernels_by_grid.txt
python tools/prof_summary.py $TT 40 --by-grid --match at::native --window-ms 250 0 > $O/train_graph_framework_kernels.txt
rm -rf $O/ttrace
head -24 $O/train_graph_kernels_by_family.txt; head -16 $O/train_graph_kernels_by_grid.txt | cut -c1-140
