#!/bin/bash
# copy what tools/prof_r06.sh left under gpurun_out/r06 (and the parity files of the last `pytest -m gpu` run) into profiles/ under their committed names
cd "$(dirname "$0")/.."
O=gpurun_out/r06; P=profiles
cat $O/bench_driver_cmd_{1,2,3,4,5}.json > $P/r06_bench_driver_command_5_fresh_processes.jsonl
cp $O/bench_native.json $P/r06_bench_native.json; cp $O/bench_native_200.json $P/r06_bench_native_200_steps.json
cp $O/bench_native_b4.json $P/r06_bench_native_batch4.json; cp $O/bench_native_b8.json $P/r06_bench_native_batch8.json
cp $O/kernel_stats.csv $P/r06_bench_native_kernel_stats.csv; cp $O/kernels_by_grid.txt $P/r06_bench_native_kernels_by_grid.txt
cp $O/bench_under_rocprof.json $P/r06_bench_native_under_rocprof.json
cp $O/bench_train.json $P/r06_bench_train.json; cp $O/bench_train_graph.json $P/r06_bench_train_graph.json
[ -f $O/bench_train_graph_b4.json ] && cp $O/bench_train_graph_b4.json $P/r06_bench_train_graph_batch4.json
cp $O/k1_bench.txt $P/r06_k1_bench.txt; cp $O/k1_bench_backward.txt $P/r06_k1_bench_backward.txt; cp $O/k1_corr_rows.txt $P/r06_k1_corr_rows.txt; cp $O/r06_k1_hbm_traffic_pmc.json $P/r06_k1_hbm_traffic_pmc.json
(cat $O/pmc_fetch_k1.txt; cat $O/pmc_write_k1.txt) > $P/r06_k1_hbm_traffic_pmc.txt
cp $O/mfma_util_b4.txt $P/r06_k3_mfma_util_pmc.txt; cp $O/sequence.jsonl $P/r06_sequence_bench.jsonl; cp $O/stress_bench.txt $P/r06_stress_bench.txt
cp $O/train_graph_kernels_by_family.txt $P/r06_train_graph_kernels_by_family.txt; cp $O/train_graph_kernels_by_grid.txt $P/r06_train_graph_kernels_by_grid.txt
python tools/parity_planted_summary.py gpurun_out > $P/r06_parity_planted.txt 2>&1
(echo '# tests/test_fuzz_gpu.py: tests/fuzz_e2e.py sweeps, launch-plan engine vs the oracle per frame / train-mode module path vs the float64 oracle'; cat gpurun_out/parity_random_geometries_inference.txt gpurun_out/parity_random_geometries_train.txt) > $P/r06_parity_random_geometries.txt
python tools/parity_stagewise_summary.py > $P/r06_parity_stagewise.txt 2>&1
cp $O/x6s_bench.txt $P/r06_x6s_bench.txt; cp $O/layer_table_b1.txt $P/r06_layer_table_batch1.txt; cp $O/layer_table_b4.txt $P/r06_layer_table_batch4.txt
cp $O/x6p_layers.txt $P/r06_x6p_layers.txt; cp $O/x6p_phase_ablation.txt $P/r06_x6p_phase_ablation.txt; cp $O/x6p_workgroup_trace.txt $P/r06_x6p_workgroup_trace.txt; cp $O/bench_x6p_ab.jsonl $P/r06_bench_x6p_on_off.jsonl
cp $O/bench_train_2ranks_one_device.json $P/r06_bench_train_2ranks_one_device.json
cp $O/train_graph_framework_kernels.txt $P/r06_train_graph_framework_kernels.txt
grep -v amdgpu.ids $O/wgrad_bench.txt > $P/r06_wgrad_bench.txt; grep -v amdgpu.ids $O/bn_small_bench.txt > $P/r06_bn_small_bench.txt
cp $O/replay_host_vs_device.txt $P/r06_train_replay_host_vs_device.txt
python - > $P/r06_parity_timed_forms.txt <<'PY2'
import json
print("# tests/test_fullsize_gpu.py::test_end_to_end_fixtures_through_the_forms_bench_times on 1xMI355X: the nine reference-made fixtures through")
print("# InferenceEngine(pipeline=3, inputs='bind') (single-frame configurations, four calls) / the two-phase begin-finish schedule (temporal ones)")
print("%-16s %5s %4s %12s %12s %10s %10s %s" % ("fixture", "frame", "call", "EPE", "EPE ref", "|dEPE|", "max |d|", "bit-identical to the plain engine"))
for r in json.load(open("gpurun_out/parity_end_to_end_timed_forms.json")):
    print("%-16s %5d %4d %12.6f %12.6f %10.2e %10.2e %s" % (r["fixture"], r["frame"], r["call"], r["epe"], r["epe_reference"], r["delta_epe"], r["max_abs"], r["bit_identical_to_plain_engine"]))
PY2
python - > $P/r06_parity_temporal_tail_audit.txt <<'PY'
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
