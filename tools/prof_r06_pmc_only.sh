set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --calibrate"
(cd /tmp && timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_write.log 2>&1)
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/k1_traffic.py $F $W > $O/r06_k1_hbm_traffic_pmc.json 2> $O/k1_traffic.err; tail -3 $O/k1_traffic.err
python tools/pmc_summary.py $F "%block_cost%" > $O/pmc_fetch_k1.txt; python tools/pmc_summary.py $W "%block_cost%" > $O/pmc_write_k1.txt
python tools/pmc_summary.py $F "%calib%" >> $O/pmc_fetch_k1.txt; python tools/pmc_summary.py $W "%calib%" >> $O/pmc_write_k1.txt
rm -rf $O/pmc_fetch $O/pmc_write
head -12 $O/r06_k1_hbm_traffic_pmc.json
