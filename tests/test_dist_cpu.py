"""CPU, world_size 2 over gloo: the data-parallel path (unit sharding, bucketed gradient averaging,
parameter broadcast) gives the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _model():
    from temporalstereo_amd.layers import Conv3d
    torch.manual_seed(0)
    return nn.Sequential(Conv3d(2, 4, (1, 3, 3), 1, (0, 1, 1), bias=False, norm=('BN3d', 4), activation='SiLU'),
                         Conv3d(4, 4, (3, 1, 1), 1, (1, 0, 0), bias=True, norm=None, activation=None),
                         Conv3d(4, 1, 1, 1, 0, bias=False, norm=None, activation=None))


def _data(n):
    g = torch.Generator().manual_seed(1)
    return torch.randn(n, 2, 3, 6, 8, generator=g), torch.randn(n, 1, 3, 6, 8, generator=g)


def _worker(rank, world, port, bucket_bytes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from temporalstereo_amd import dist as tsd
    r, w, _ = tsd.init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    if rank != 0:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)                 # diverge on purpose; broadcast must repair it
    tsd.broadcast_parameters(model)
    model[0].norm.eval()                     # per-sample independent so that averaging is exact
    x, y = _data(4)
    mine = tsd.shard_units(4, rank, world)
    gb = tsd.GradientBuckets(model.parameters(), bucket_bytes=bucket_bytes)
    assert len(gb.buckets) >= (2 if bucket_bytes < 200 else 1)
    loss = ((model(x[mine]) - y[mine]) ** 2).mean()
    loss.backward()
    gb.finish()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [64, 1 << 20])
def test_bucketed_gradient_averaging_matches_single_process(tmp_path, bucket_bytes):
    out = str(tmp_path / "grads.pt")
    mp.spawn(_worker, args=(2, _free_port(), bucket_bytes, out), nprocs=2, join=True)
    got = torch.load(out)
    model = _model()
    model[0].norm.eval()
    x, y = _data(4)
    # mean over ranks of per-rank mean losses == mean over all 4 samples (equal shard sizes)
    ((model(x) - y) ** 2).mean().backward()
    for k, p in model.named_parameters():
        torch.testing.assert_close(got[k], p.grad, rtol=1e-5, atol=1e-6)


def test_shard_units():
    from temporalstereo_amd.dist import shard_units
    assert shard_units(10, 0, 4) == [0, 4, 8] and shard_units(10, 3, 4) == [3, 7]
    assert shard_units(10, 1, 4, drop_last=True) == [1, 5]
    assert sorted(sum((shard_units(7, r, 3) for r in range(3)), [])) == list(range(7))


def _unused_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from temporalstereo_amd import dist as tsd
    tsd.init_distributed("gloo")
    model = _model()
    model.phi = nn.Parameter(torch.zeros(1))          # declared, never used (FineAggregation.phi, fine.py:34)
    model[0].norm.eval()
    tsd.broadcast_parameters(model)
    x, y = _data(4)
    mine = tsd.shard_units(4, rank, world)
    gb = tsd.GradientBuckets(model.parameters(), bucket_bytes=1 << 20)
    launched = []
    for step in range(2):
        model.zero_grad(set_to_none=True)
        ((model(x[mine]) - y[mine]) ** 2).mean().backward()
        launched.append(gb.launched_in_backward)
        gb.finish()
        assert model.phi.grad is None                 # outside the graph on every rank: stays None, as in the reference
    # step 0 cannot fill the bucket (phi never fires); from step 1 on the all-reduce goes out during backward
    assert launched == [0, 1], launched
    with pytest.raises(RuntimeError):                 # a second backward without finish() is refused, not silently dropped
        ((model(x[mine]) - y[mine]) ** 2).mean().backward()
        ((model(x[mine]) - y[mine]) ** 2).mean().backward()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_buckets_drop_unused_parameters_and_overlap_from_the_second_step(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_unused_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert "phi" not in torch.load(out)
