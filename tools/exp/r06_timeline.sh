# experiment: timeline of one pass (one at a time) with and without the ping-pong x6 form: gaps around its launches
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06tl
rm -rf $O; mkdir -p $O
for v in 0 1; do
  (cd /tmp && TS_X6P=$v timeout 300 rocprofv3 --kernel-trace -d $O/tr$v -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --frames-in-flight 1 --steps 30 --warmup 5 --no-cpu-baseline --no-extras ${TL_ARGS} > /dev/null" > $O/tr$v.log 2>&1)
  T=$(find $O/tr$v -name "*.db" | head -1)
  python tools/exp/pass_timeline.py $T 25 > $O/timeline_x6p$v.txt
  rm -rf $O/tr$v
done
