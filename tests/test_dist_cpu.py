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


def _paused_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from temporalstereo_amd import dist as tsd
    tsd.init_distributed("gloo")
    model = _model()
    model.side = nn.Parameter(torch.ones(1))          # receives gradient ONLY in the paused() micro-batch
    model[0].norm.eval()
    tsd.broadcast_parameters(model)
    x, y = _data(8)
    mine = tsd.shard_units(8, rank, world)
    a, b = mine[:2], mine[2:]
    gb = tsd.GradientBuckets(model.parameters(), bucket_bytes=64)
    for step in range(2):                             # the first step (participation agreed) and a steady-state one
        model.zero_grad(set_to_none=True)
        with gb.paused():
            (((model(x[a]) - y[a]) ** 2).mean() * 0.5 + (model.side * (rank + 1.0)).sum()).backward()
        (((model(x[b]) - y[b]) ** 2).mean() * 0.5).backward()
        gb.finish()
    if rank == 0:
        torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulated_under_paused_is_kept(tmp_path):
    """A parameter whose only gradient of the step arrived under gb.paused() (accumulation micro-batch) is averaged, not zeroed."""
    out = str(tmp_path / "g.pt")
    mp.spawn(_paused_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = _model()
    model[0].norm.eval()
    x, y = _data(8)
    ((model(x) - y) ** 2).mean().backward()            # == mean over ranks of (0.5 mean(a) + 0.5 mean(b))
    for k, p in model.named_parameters():
        torch.testing.assert_close(got[k], p.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got["side"], torch.tensor([1.5]))      # mean of d/d side = rank + 1 over the two ranks


# ------------------------------------------------------------------------------------------------ the real aggregator, ws 2
def _tiny_aggregator(setattr_=setattr):
    """The registered TEMPORALSTEREO module at the tiny fixture dimensions, fp64, with the two GPU-only ops of its forward bound
    to the oracle (tests may use it): the module graph, its BatchNorm layers and the data-parallel wiring are what is under test."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle
    from helpers import load, dims_from_golden, synth_state, aggregator_inputs
    import temporalstereo_amd as ts
    from temporalstereo_amd import functional as TF
    setattr_(TF, "block_cost", oracle.block_cost)              # (the test process passes monkeypatch.setattr: undone afterwards)
    setattr_(TF, "topk_softargmax", oracle.topk_softargmax)
    g = load("agg_tiny_single")
    dims = dims_from_golden(g)
    net = ts.TEMPORALSTEREO(
        coarse=ts.CoarseAggregation(dims['coarse']['in_planes'], dims['coarse']['C'], dims['coarse']['num_sample']),
        fine=ts.FineAggregation(dims['fine']['in_planes'], dims['fine']['C'], 5),
        precise=ts.PreciseAggregation(dims['precise']['in_planes'], dims['precise']['C'], 5))
    net.load_state_dict(synth_state(dims, int(g["seed"]), golden=g), strict=True)
    lf, rf, il, ir, _ = aggregator_inputs(g, dims)
    d = lambda x: x.double()
    return net.double().train(), ([d(x) for x in lf], [d(x) for x in rf], d(il), d(ir))


def _objective(out):
    disps, costs, samples, offs = out[0], out[1], out[2], out[3]
    return sum(x.mean() for x in disps) + sum((c * 0.01).tanh().mean() for c in costs) + sum(o.mean() for o in offs)


def _train_two_steps(net, inputs, buckets=None):
    params = [p for p in net.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        _objective(net(*inputs, {})).backward()
        if buckets is not None:
            buckets.finish()
        opt.step()
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def _agg_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from temporalstereo_amd import dist as tsd
    tsd.init_distributed("gloo")
    net, (lf, rf, il, ir) = _tiny_aggregator()
    net = tsd.sync_batchnorm(net)
    assert sum(isinstance(m, tsd.SyncBatchNorm) for m in net.modules()) > 80
    tsd.broadcast_parameters(net)
    mine = slice(rank, rank + 1)                           # the fixture's batch of 2: one pair per rank
    inputs = ([x[mine] for x in lf], [x[mine] for x in rf], il[mine], ir[mine])
    gb = tsd.GradientBuckets(net.parameters(), bucket_bytes=256 << 10)
    sd = _train_two_steps(net, inputs, gb)
    assert gb.launched_in_backward > 0                     # second step: buckets went out during backward
    if rank == 0:
        torch.save(sd, out)
    dist.barrier()
    dist.destroy_process_group()


def test_real_aggregator_two_ranks_sync_batchnorm_matches_single_process(tmp_path, monkeypatch):
    """Two optimisation steps of the tiny REAL aggregator, one pair per rank, SyncBatchNorm + bucketed gradient averaging ==
    the same two steps in one process on the batch of two with plain BatchNorm (parameters AND running statistics)."""
    out = str(tmp_path / "sd.pt")
    mp.spawn(_agg_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    net, inputs = _tiny_aggregator(monkeypatch.setattr)
    want = _train_two_steps(net, inputs)
    assert set(got) == set(want)
    for k in want:
        if k.endswith("num_batches_tracked"):
            assert int(got[k]) == int(want[k]), k
            continue
        torch.testing.assert_close(got[k], want[k], rtol=1e-7, atol=1e-9, msg=lambda m, k=k: "%s: %s" % (k, m))


def _branch_worker(rank, world, port, out):
    """A parameter that takes part on one rank only from the second step on (data-dependent branch): the buckets must still go
    out in the same order on both ranks, the absent rank contributing zeros."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from temporalstereo_amd import dist as tsd
    tsd.init_distributed("gloo")
    torch.manual_seed(0)
    a, b, c = (nn.Parameter(torch.randn(4, 4)) for _ in range(3))
    gb = tsd.GradientBuckets([a, b, c], bucket_bytes=16)          # one bucket per parameter
    x = torch.ones(2, 4) * (rank + 1)
    for step in range(3):
        for p in (a, b, c):
            p.grad = None
        y = x @ a + x @ c
        if step == 0 or rank == 0:                                   # `b` is used everywhere on the first step, then on rank 0 only
            y = y + x @ b
        y.sum().backward()
        gb.finish()
    if rank == 0:
        torch.save({"a": a.grad.clone(), "b": b.grad.clone(), "c": c.grad.clone()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_buckets_keep_one_order_when_a_parameter_fires_on_one_rank_only(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_branch_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # d(sum(x @ p))/dp = column sums of x repeated: rank r contributes 2 * (r + 1) per entry; averaged over 2 ranks
    assert torch.allclose(got["a"], torch.full((4, 4), (2 * 1 + 2 * 2) / 2.0))
    assert torch.allclose(got["c"], torch.full((4, 4), (2 * 1 + 2 * 2) / 2.0))
    assert torch.allclose(got["b"], torch.full((4, 4), (2 * 1 + 0.0) / 2.0))      # rank 1 did not use it: zeros, not a stale gradient
