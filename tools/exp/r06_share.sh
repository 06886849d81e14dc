cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06share; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $O/t -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras > $O/bench.json" > $O/trace.log 2>&1)
T=$(find $O/t -name "*.db" | head -1)
python -c "
import sqlite3,sys
db=sqlite3.connect('$T'); print([r[1] for r in db.execute('pragma table_info(kernels)')])"
python tools/timeline_share.py $T 60 --window-ms 40 2 > $O/share_pipelined.txt 2>&1
head -70 $O/share_pipelined.txt
rm -rf $O/t
