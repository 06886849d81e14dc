cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${OUT:-r02t}
rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --mode ${TRAIN_MODE:-train} --steps ${TRAIN_STEPS:-10} --warmup 3 > $O/bench_train_under_rocprof.json" > $O/trace.log 2>&1)
T=$(find $O/trace -name "*.db" | head -1)
python tools/prof_summary.py $T 60 --by-grid --window-ms ${WINDOW_MS:-250} 0 > $O/train_kernels_by_grid.txt; python tools/prof_summary.py $T --by-family --window-ms ${WINDOW_MS:-250} 0 > $O/train_kernels_by_family.txt
python tools/prof_summary.py $T 40 --by-grid --match ${MATCH:-at::native} --window-ms ${WINDOW_MS:-250} 0 > $O/train_framework_kernels.txt
rm -rf $O/trace
cat $O/train_kernels_by_family.txt; cat $O/bench_train_under_rocprof.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
