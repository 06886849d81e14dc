#!/bin/bash
# copy what tools/prof_r04.sh left under gpurun_out/r04 (and the parity files of the last `pytest -m gpu` run) into profiles/ under their committed names
cd "$(dirname "$0")/.."
O=gpurun_out/r04; P=profiles
cat $O/bench_driver_cmd_{1,2,3,4,5}.json > $P/r04_bench_driver_command_5_fresh_processes.jsonl
cp $O/bench_native.json $P/r04_bench_native.json; cp $O/bench_native_200.json $P/r04_bench_native_200_steps.json
cp $O/bench_native_b4.json $P/r04_bench_native_batch4.json; cp $O/bench_native_b8.json $P/r04_bench_native_batch8.json
cp $O/kernel_stats.csv $P/r04_bench_native_kernel_stats.csv; cp $O/kernels_by_grid.txt $P/r04_bench_native_kernels_by_grid.txt
cp $O/bench_under_rocprof.json $P/r04_bench_native_under_rocprof.json
cp $O/bench_train.json $P/r04_bench_train.json; cp $O/bench_train_graph.json $P/r04_bench_train_graph.json
cp $O/k1_bench.txt $P/r04_k1_bench.txt; cp $O/r04_k1_hbm_traffic_pmc.json $P/r04_k1_hbm_traffic_pmc.json
(cat $O/pmc_fetch_k1.txt; cat $O/pmc_write_k1.txt) > $P/r04_k1_hbm_traffic_pmc.txt
cp $O/mfma_util_b4.txt $P/r04_k3_mfma_util_pmc.txt; cp $O/sequence.jsonl $P/r04_sequence_bench.jsonl; cp $O/stress_bench.txt $P/r04_stress_bench.txt
cp $O/train_graph_kernels_by_family.txt $P/r04_train_graph_kernels_by_family.txt; cp $O/train_graph_kernels_by_grid.txt $P/r04_train_graph_kernels_by_grid.txt
python tools/parity_planted_summary.py gpurun_out > $P/r04_parity_planted.txt 2>&1
python tools/parity_stagewise_summary.py > $P/r04_parity_stagewise.txt 2>&1
cp $O/x6s_bench.txt $P/r04_x6s_bench.txt; cp $O/layer_table_b1.txt $P/r04_layer_table_batch1.txt; cp $O/layer_table_b4.txt $P/r04_layer_table_batch4.txt
cp $O/bench_train_2ranks_one_device.json $P/r04_bench_train_2ranks_one_device.json
python - > $P/r04_parity_temporal_tail_audit.txt <<'PY'
import json
print("# tests/test_fullsize_gpu.py::test_temporal_tail_is_explained_pixel_by_pixel on 1xMI355X: free-running sequences against the oracle,")
print("# every moved pixel (> 1e-3 px at its level) explained: near-tie of the oracle / sort near-tie / reach of such an event / inherited from the")
print("# level above or from the entering state / within the soft-argmax's sensitivity to the measured cost error.  unexplained must be 0.")
print("%-12s %5s %-8s %8s %7s %10s %12s %10s" % ("config", "frame", "level", "pixels", "moved", "near-ties", "unexplained", "max move"))
for r in json.load(open("gpurun_out/parity_temporal_tail_audit.json")):
    c = r["config"][:10]
    if "level" in r:
        print("%-12s %5d %-8s %8d %7d %10d %12d %10.2e" % (c, r["frame"], r["level"], r["pixels"], r["moved"], r["near_ties"], r["unexplained"], r["max_move"]))
    elif "off_by_0p05" in r:
        print("%-12s %5d %-8s full-resolution pixels off by > 0.05 px: %d, outside every explained 1/4-resolution pixel: %d" % (c, r["frame"], "full", r["off_by_0p05"], r["uncovered"]))
    elif "differing" in r:
        print("%-12s %5d %-8s entering state differs by > 1e-4 at %d of %d memory pixels" % (c, r["frame"], "state", r["differing"], r["pixels"]))
PY
