"""GPU, two ranks: the peer-mailbox exchange of csrc/peer.hip (SyncBatchNorm's statistics exchanges as kernels over hipIpc-mapped
memory, reference role: torch's SyncBatchNorm under Lightning's sync_batchnorm=True, projects/TemporalStereo/dist_train.py:94).
Both ranks on device 0 (a single-GPU box), handles over gloo; the mapping, the slot protocol and the graph replay are the
production path -- what the box cannot show is the xGMI hop between two devices."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _run(tmp_path, mode, world=2):
    env = dict(os.environ, TS_BENCH_BACKEND="gloo", TS_BENCH_DEVICE="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
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
    """One rank never issues its exchange: the other's kernel gives up after its bound (~2 s of wall clock, csrc/peer.hip) and
    PeerGroup.check() raises on both ranks (the outcome is agreed on through the process group)."""
    ranks = _run(tmp_path, "missing_peer")
    assert [int(r["raised"]) for r in ranks] == [1, 1]
