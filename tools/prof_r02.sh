set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02
rm -rf $O; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --calibrate"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_write.log 2>&1)
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/k1_traffic.py $F $W > $O/r02_k1_hbm_traffic_pmc.json 2> $O/k1_traffic.err; tail -3 $O/k1_traffic.err
python tools/pmc_summary.py $F "%block_cost%" > $O/pmc_fetch_k1.txt; python tools/pmc_summary.py $W "%block_cost%" > $O/pmc_write_k1.txt
python tools/pmc_summary.py $F "%calib%" >> $O/pmc_fetch_k1.txt; python tools/pmc_summary.py $W "%calib%" >> $O/pmc_write_k1.txt
# kernel trace + stats of the bench command
(cd /tmp && rocprofv3 --kernel-trace --stats -d $O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json" > $O/trace.log 2>&1)
T=$(find $O/trace -name "*.db" | head -1)
python tools/prof_summary.py $T 200 --by-grid > $O/kernels_by_grid.txt
python tools/prof_summary.py $T 300 > $O/kernel_stats.txt
# MFMA utilisation at batch 4
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $O/mfma -o m -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --batch 4" > $O/mfma.log 2>&1)
M=$(find $O/mfma -name "*.db" | head -1)
python tools/mfma_util.py $M > $O/mfma_util_b4.txt
# the bench line itself, with the all-host-threads CPU figure once
python bench.py --cpu-all-cores > $O/bench_native.json 2> $O/bench_native.err
python tools/k1_bench.py > $O/k1_bench.txt 2>&1
python tools/sequence_bench.py > $O/sequence.jsonl 2>/dev/null; python tools/sequence_bench.py --one-at-a-time > $O/sequence_one_at_a_time.jsonl 2>/dev/null; python tools/sequence_bench.py --configs 3 --frames 4 >> $O/sequence.jsonl 2>/dev/null; python bench.py --mode train --steps 20 --warmup 5 > $O/bench_train.json 2>/dev/null; python bench.py --batch 4 --no-cpu-baseline --no-extras > $O/bench_native_b4.json 2>/dev/null; python bench.py --batch 8 --no-cpu-baseline --no-extras > $O/bench_native_b8.json 2>/dev/null; python tools/parity_report.py --seeds 8 2 1 2 > $O/parity_end_to_end.txt 2>/dev/null
rm -rf $O/pmc_fetch $O/pmc_write $O/trace $O/mfma
ls -la $O
