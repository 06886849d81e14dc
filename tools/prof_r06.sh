# (every profiler call is bounded: a rocprofv3 that does not come back must not eat the GPU call -- one did in round 6, 15 minutes)
# round 6: everything the numbers in DESIGN.md / README.md / profiles/ come from, one GPU call (see profiles/README.md)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
rm -rf $O; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --calibrate"
(cd /tmp && timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_fetch.log 2>&1)
(cd /tmp && timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w -- bash -c "cd $GRAFT_REPO_ROOT && $B" > $O/pmc_write.log 2>&1)
F=$(find $O/pmc_fetch -name "*.db" | head -1); W=$(find $O/pmc_write -name "*.db" | head -1)
python tools/k1_traffic.py $F $W > $O/r06_k1_hbm_traffic_pmc.json 2> $O/k1_traffic.err; tail -3 $O/k1_traffic.err
python tools/pmc_summary.py $F "%block_cost%" > $O/pmc_fetch_k1.txt; python tools/pmc_summary.py $W "%block_cost%" > $O/pmc_write_k1.txt
python tools/pmc_summary.py $F "%calib%" >> $O/pmc_fetch_k1.txt; python tools/pmc_summary.py $W "%calib%" >> $O/pmc_write_k1.txt
# kernel trace + stats of the DRIVER's bench command (20 steps / 5 warm-up)
(cd /tmp && timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json" > $O/trace.log 2>&1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" $O/kernel_stats.csv
(cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $O/trace2 -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --no-cpu-baseline --no-extras > /dev/null" > $O/trace2.log 2>&1)
T=$(find $O/trace2 -name "*.db" | head -1)
python tools/prof_summary.py $T 200 --by-grid > $O/kernels_by_grid.txt
# MFMA utilisation at batch 4
(cd /tmp && timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace -d $O/mfma -o m -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --batch 4" > $O/mfma.log 2>&1)
M=$(find $O/mfma -name "*.db" | head -1)
python tools/mfma_util.py $M > $O/mfma_util_b4.txt
# the training step under the profiler (replayed), by family and by grid
(cd /tmp && timeout 420 rocprofv3 --kernel-trace -d $O/ttrace -o t -- bash -c "cd $GRAFT_REPO_ROOT && python bench.py --mode train-graph --steps 30 --warmup 5 > $O/bench_train_graph_under_rocprof.json" > $O/ttrace.log 2>&1)
TT=$(find $O/ttrace -name "*.db" | head -1)
python tools/prof_summary.py $TT --by-family --window-ms 250 0 > $O/train_graph_kernels_by_family.txt
python tools/prof_summary.py $TT 80 --by-grid --window-ms 250 0 > $O/train_graph_kernels_by_grid.txt
python tools/prof_summary.py $TT 40 --by-grid --match at::native --window-ms 250 0 > $O/train_graph_framework_kernels.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/trace $O/trace2 $O/mfma $O/ttrace
# the bench lines themselves: the driver's command five times (fresh processes), the long form, batches, training, sequences
for i in 1 2 3 4 5; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_driver_cmd_$i.json 2>/dev/null; done
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-all-cores > $O/bench_native.json 2> $O/bench_native.err
python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/bench_native_200.json 2>/dev/null
python bench.py --batch 4 --no-cpu-baseline --no-extras > $O/bench_native_b4.json 2>/dev/null; python bench.py --batch 8 --no-cpu-baseline --no-extras > $O/bench_native_b8.json 2>/dev/null
python bench.py --mode train --steps 30 --warmup 5 > $O/bench_train.json 2>/dev/null; python bench.py --mode train-graph --steps 30 --warmup 5 > $O/bench_train_graph.json 2>/dev/null
python bench.py --mode train-graph --batch 4 --steps 20 --warmup 4 > $O/bench_train_graph_b4.json 2>/dev/null      # BASELINE configs[2]: T=2, batch 4
python tools/k1_bench.py > $O/k1_bench.txt 2>&1; python tools/k1_bench.py --backward 2>&1 | grep '^K1' > $O/k1_bench_backward.txt
python tools/exp/k1_corr_ablate.py 2>&1 | grep -v amdgpu.ids > $O/k1_corr_rows.txt
python tools/stress_bench.py > $O/stress_bench.txt 2>&1
python tools/x6s_bench.py > $O/x6s_bench.txt 2>&1
# the ping-pong x6 form (csrc/conv_x6p.hip): layer table with the form off / on (default rule) / forced on 8- and 4-row half tiles, its fp64
# accuracy sweep forced on every grid, the phase ablation and the cycle stamps of one workgroup
(echo "## TS_X6P=0 (ig_conv_x6_kernel)"; TS_X6P=0 python tools/exp/x6p_check.py --child time; echo "## default rule"; python tools/exp/x6p_check.py --child time;
 echo "## TS_X6P_MIN_WGS=1 TS_X6P_HR=8"; TS_X6P_MIN_WGS=1 TS_X6P_HR=8 python tools/exp/x6p_check.py --child time;
 echo "## TS_X6P_MIN_WGS=1 TS_X6P_HR=4"; TS_X6P_MIN_WGS=1 TS_X6P_HR=4 python tools/exp/x6p_check.py --child time;
 echo "## accuracy vs an fp64 convolution, TS_X6P_MIN_WGS=1 TS_X6P_HR=8"; TS_X6P_MIN_WGS=1 TS_X6P_HR=8 python tools/exp/x6p_check.py --child acc;
 echo "## accuracy vs an fp64 convolution, TS_X6P_MIN_WGS=1 TS_X6P_HR=4"; TS_X6P_MIN_WGS=1 TS_X6P_HR=4 python tools/exp/x6p_check.py --child acc) 2>&1 | grep -v amdgpu.ids > $O/x6p_layers.txt
python tools/exp/x6p_abl.py 0 1 2 4 8 16 3 6 7 29 31 2>&1 | grep -v amdgpu.ids > $O/x6p_phase_ablation.txt
(TS_X6P_MIN_WGS=1 TS_X6P_HR=8 TS_X6P_TRACE=3 python tools/exp/x6p_trace.py; TS_X6P_MIN_WGS=1 TS_X6P_HR=8 TS_X6P_DBG=29 TS_X6P_TRACE=3 python tools/exp/x6p_trace.py;
 TS_X6P_MIN_WGS=1 TS_X6P_HR=4 TS_X6P_TRACE=3 python tools/exp/x6p_trace.py 1 128 32 136 240; TS_X6P_MIN_WGS=1 TS_X6P_HR=8 TS_X6P_TRACE=-2 python tools/exp/x6p_wgtimes.py) 2>&1 | grep -v amdgpu.ids > $O/x6p_workgroup_trace.txt
for e in "TS_X6P=0" "TS_X6P=1"; do echo "## $e"; env $e python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | tail -1; env $e python bench.py --batch 4 --steps 30 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1; done > $O/bench_x6p_ab.jsonl
python tools/exp/wgrad_bench.py > $O/wgrad_bench.txt 2>&1; python tools/exp/bn_small_bench.py > $O/bn_small_bench.txt 2>&1
python tools/exp/replay_host_vs_device.py 2>&1 | tail -3 > $O/replay_host_vs_device.txt
python tools/layer_table.py > $O/layer_table_b1.txt 2>&1; python tools/layer_table.py --batch 4 > $O/layer_table_b4.txt 2>&1
TS_BENCH_BACKEND=gloo TS_BENCH_DEVICE=0 timeout 600 python bench.py --mode train --gpus 2 --steps 8 --warmup 3 2>/dev/null | grep '^{' > $O/bench_train_2ranks_one_device.json
python tools/sequence_bench.py > $O/sequence.jsonl 2>/dev/null; python tools/sequence_bench.py --configs 3 --frames 4 >> $O/sequence.jsonl 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/bench_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(r["value"], 1), r.get("one_pass_at_a_time", {}).get("value"), r.get("f32_mfma_only", {}).get("value"), (r.get("roofline") or {}).get("frac"),
              (r.get("training") or {}).get("ms_per_step"), ((r.get("training") or {}).get("hipgraph") or {}).get("ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
ls -la $O
