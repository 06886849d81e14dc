# kernel trace of the last (one-pass-at-a-time) phase of the default bench run: per-kernel durations without overlap
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02p; rm -rf $O; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace -d $O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --no-extras --batch ${BATCH:-1} > $O/bench.json" > $O/trace.log 2>&1)
T=$(find $O/trace -name "*.db" | head -1)
python tools/prof_summary.py $T 70 --by-grid --window-ms ${WINDOW_MS:-12} 0 > $O/kernels_by_grid.txt
python tools/prof_summary.py $T --by-family --window-ms ${WINDOW_MS:-12} 0 > $O/kernels_by_family.txt
rm -rf $O/trace
head -75 $O/kernels_by_grid.txt | cut -c1-140
