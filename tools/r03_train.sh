#!/bin/bash
# round 3, first GPU call: train the planted-scene checkpoint, look at end-to-end parity with it, and time the driver's exact
# bench command cold (with / without the conditioning phase)
set -x
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/train_checkpoint.py --steps ${STEPS:-1500} --budget-s ${BUDGET:-900} --out gpurun_out/ckpt_planted.npz > gpurun_out/train_ckpt.log 2>&1
tail -5 gpurun_out/train_ckpt.log
python tools/parity_planted.py --ckpt gpurun_out/ckpt_planted.npz --seeds 2 --configs 1,2 --fp64 > gpurun_out/parity_planted.log 2>&1
tail -12 gpurun_out/parity_planted.log
for i in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --condition-s 0 > gpurun_out/bench_cold_nocond_$i.json 2> gpurun_out/bench_cold_nocond_$i.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_cold_cond_$i.json 2> gpurun_out/bench_cold_cond_$i.err
done
python bench.py --gpus 1 --steps 200 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/bench_200.json 2> gpurun_out/bench_200.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print(f, r["value"], r.get("conditioning"), r.get("one_pass_at_a_time",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
