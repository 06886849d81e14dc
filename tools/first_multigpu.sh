#!/bin/bash
# The FIRST run on a multi-GPU MI355X node, in the order in which a failure is cheapest to read.
# Nothing here has ever crossed a device boundary: RCCL with more than one rank, hipIpc mappings of the peer mailboxes over xGMI,
# system-scope fences between devices.  Every stage is bounded by `timeout`; a stage that fails stops the script with its log.
#   usage: bash tools/first_multigpu.sh [N]        (N ranks / GPUs, default: all visible devices, at most 8)
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}      # the host driver supports dmabuf IPC only
N=${1:-$(python -c "import torch; print(min(8, torch.cuda.device_count()))")}
O=gpurun_out/first_multigpu; mkdir -p $O
echo "== devices: $N =="; [ "$N" -ge 2 ] || { echo "needs at least two devices"; exit 2; }
stage() { echo; echo "== $1 =="; shift; if ! timeout 900 "$@" > $O/stage.log 2>&1; then tail -40 $O/stage.log; echo "FAILED: $*"; exit 1; fi; tail -5 $O/stage.log; }

# 1. the mailboxes alone: mapped across devices (hipIpc over xGMI), gather / reduce bit-exact, graph replay, a missing peer times out
#    (tests/helpers.multi_rank_env picks one rank per device + RCCL by itself when the box has the devices)
stage "peer mailboxes across devices" python -m pytest tests/test_peer_gpu.py -x -q -s
# 2. two ranks of the real aggregator: SyncBatchNorm over RCCL, buckets during backward; then the same statistics through the mailboxes,
#    eager and replayed from a hipGraph
stage "two-rank data-parallel step (collectives, peer, peer + hipGraph)" python -m pytest tests/test_ddp_gpu.py -x -q -s
# 3. the driver's launch of the inference bench, as the driver issues it
PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
stage "bench.py --gpus $N (replicas, the driver's command)" python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $N --steps 20 --warmup 5 --no-extras
grep '^{' $O/stage.log | tail -1 > $O/bench_replicas_$N.json
# 4. the training step on N ranks: collectives leg first (its line is printed before the peer legs start), then peer / peer + hipGraph;
#    a mailbox that cannot be mapped or a peer that does not answer ends that leg with the reason in the JSON, not the run
stage "bench.py --mode train --gpus $N (collectives, then peer legs)" python bench.py --mode train --gpus $N --steps 10 --warmup 4
grep '^{' $O/stage.log | tail -1 > $O/bench_train_$N.json
python - <<PY
import json
r = json.load(open("$O/bench_replicas_$N.json")); t = json.load(open("$O/bench_train_$N.json"))["training"]
print("replicas x$N: %.0f pairs/s" % r["value"])
print("training dp$N: collectives %.2f ms/step;" % t["ms_per_step"], "peer:", t.get("peer"), "; peer + hipGraph:", t.get("peer_hipgraph"))
PY
echo "all stages passed; outputs under $O/"
