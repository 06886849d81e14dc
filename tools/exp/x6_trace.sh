cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02x; rm -rf $O; mkdir -p $O
for v in 1 0; do
(cd /tmp && TS_CONV_X6=$v rocprofv3 --kernel-trace -d $O/trace$v -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --no-extras --batch ${BATCH:-4} --steps 20 --warmup 5 > $O/bench$v.json" > $O/trace$v.log 2>&1)
T=$(find $O/trace$v -name "*.db" | head -1)
python tools/prof_summary.py $T 40 --by-grid --window-ms 40 0 > $O/kernels_x6_$v.txt
rm -rf $O/trace$v
done
head -45 $O/kernels_x6_1.txt | cut -c1-150
