# experiment: busy time per hardware queue of the pipelined bench (six stage streams)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06qb
rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/tr -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras ${QB_ARGS} > $O/bench.json" > $O/tr.log 2>&1)
T=$(find $O/tr -name "*.db" | head -1)
python tools/exp/queue_balance.py $T 30 > $O/queue_balance.txt
rm -rf $O/tr
