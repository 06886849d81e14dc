"""GPU: the one-line JSON contract of bench.py (driver-facing) on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "pairs/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - d["config"]["batch_per_gpu"]) < 1e-6 * d["value"]      # value = pairs / time
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 1.0 and r["traffic"] is not None
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    par = d["parity"]
    assert par["delta_epe_px"] < par["tolerance_px"] == 1e-3
    # planted scene + trained checkpoint: a real ground truth, sub-pixel EPE on both sides, no tail of flipped pixels
    assert par["ground_truth"].startswith("planted") and 0.01 < par["epe_px"] < 1.0 and abs(par["epe_px"] - par["epe_reference_px"]) < 1e-3
    assert par["max_abs_diff_px"] < 0.05 and par["frac_pixels_off_by_0p01"] < 1e-4
    assert d["config"]["frames_in_flight"] == 3 and d["one_pass_at_a_time"]["value"] > 0
    # protocol: the untimed conditioning phase is bounded and reported; the x6 engine is not slower than the f32-only one it is compared with
    cond = d["conditioning"]
    assert cond["passes"] >= 10 and cond["seconds"] <= cond["bound_s"] + 0.5
    assert d["value"] > 0.95 * d["f32_mfma_only"]["value"]
    tr = d["training"]
    assert "error" not in tr and tr["ms_per_step"] > 0 and tr["hipgraph"]["ms_per_step"] < tr["ms_per_step"]


def test_driver_launch_of_eight_ranks():
    """The driver's multi-GPU command, `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 --steps K --warmup W`, end to end: on an eight-GPU node exactly as the driver runs it (one rank per device, RCCL);
    on a smaller box eight ranks on device 0 with gloo as the rendezvous / barrier / max-over-ranks backend (RCCL refuses two ranks
    on one device; TS_BENCH_BACKEND / TS_BENCH_DEVICE are the test hooks, tests/helpers.multi_rank_env).  What this exercises before the first real SCALE run: rendezvous on 127.0.0.1, every rank building its engine, the
    barrier + synchronize bracket, the MAX over ranks, ONE JSON line from rank 0 with the whole-job value, a clean exit of all ranks."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    from helpers import multi_rank_env
    env, arrangement = multi_rank_env(8)            # eight devices: the driver's command as it is (RCCL, one rank per GPU); else device 0 + gloo
    print("arrangement:", arrangement)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2", "--no-extras"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "replicas x8" and d["value"] > 0
    # whole-job aggregate: 8 ranks x batch x steps over the slowest rank's time
    assert abs(d["value"] - 8 * d["config"]["batch_per_gpu"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
