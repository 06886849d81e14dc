cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05
mkdir -p $O
(cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $O/ttrace4 -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --mode train-graph --batch 4 --steps 16 --warmup 4 > $O/bench_train_graph_b4_under_rocprof.json" > $O/ttrace4.log 2>&1)
TT=$(find $O/ttrace4 -name "*.db" | head -1)
python tools/prof_summary.py $TT --by-family --window-ms 250 0 > $O/train_graph_b4_kernels_by_family.txt
python tools/prof_summary.py $TT 40 --by-grid --window-ms 250 0 > $O/train_graph_b4_kernels_by_grid.txt
rm -rf $O/ttrace4
head -30 $O/train_graph_b4_kernels_by_family.txt
