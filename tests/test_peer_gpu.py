"""GPU, two ranks: the peer-mailbox exchange of csrc/peer.hip (SyncBatchNorm's statistics exchanges as kernels over hipIpc-mapped
memory, reference role: torch's SyncBatchNorm under Lightning's sync_batchnorm=True, projects/TemporalStereo/dist_train.py:94).
The arrangement follows the box (tests/helpers.multi_rank_env): with as many devices as ranks every rank takes its own device and
the handles travel over RCCL -- the mailboxes are then mapped ACROSS devices (hipIpc over xGMI); on a single-GPU box all ranks share
device 0 and gloo carries the handles (the mapping, the slot protocol and the graph replay are the production path either way)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import multi_rank_env

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(tmp_path, mode, world=2, **extra):
    env, arrangement = multi_rank_env(world, **extra)
    print("arrangement:", arrangement)
    out = str(tmp_path / "peer")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "peer_gpu_worker.py"), out, mode]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return [np.load(out + ".rank%d.npz" % k) for k in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_peer_all_gather_and_all_reduce(tmp_path, world):
    ranks = _run(tmp_path, "eager", world)
    for it in range(10):
        srcs = [r["src%d" % it] for r in ranks]
        want = np.stack(srcs)
        total = srcs[0].copy()
        for s in srcs[1:]:
            total = total + s                         # rank order, fp32: bit-identical on every rank
        if it % 2:
            total = total * np.float32(0.5)
        for r in ranks:
            assert np.array_equal(r["gather%d" % it], want)
            assert np.array_equal(r["sum%d" % it], total)


def test_peer_exchange_replayed_from_a_graph(tmp_path):
    ranks = _run(tmp_path, "graph")
    for k in range(5):
        want = np.stack([np.full(65, 10.0 * k + r, dtype=np.float32) for r in range(2)])
        for r in ranks:
            assert np.array_equal(r["ggather%d" % k], want)
            assert np.array_equal(r["gsum%d" % k], want[0] + want[1])


def test_a_missing_peer_times_out_instead_of_hanging(tmp_path):
    """One rank never issues its exchange: the other's kernel gives up after its bound (set to 2 s here; the default is a
    watchdog-sized 120 s, csrc/peer.hip) and PeerGroup.check() raises on both ranks (the outcome is agreed on through the process
    group); poll() -- what TrainStep asks every step, without synchronising -- reports it one poll late; reset() revives the group."""
    ranks = _run(tmp_path, "missing_peer", TS_PEER_TIMEOUT_MS=2000)
    assert [int(r["raised"]) for r in ranks] == [1, 1]
    assert int(ranks[0]["polled"]) == 1                   # the rank whose kernel timed out hears of it from poll() as well
    for r in ranks:
        assert int(r["after_reset_ok"]) == 1
