"""Data-parallel scaling of the hot path: one process per GPU, `torch.distributed` over RCCL/xGMI.

The reference delegates this to Lightning (`pl.Trainer(strategy='ddp', sync_batchnorm=True)`,
projects/TemporalStereo/dist_train.py:82-96; DistributedSampler over stereo pairs,
TemporalStereo.py:56-58).  The path shards naturally: stereo pairs (or whole temporal sequences, since
`prev_info` chains the frames of one sequence) are independent units.

  * inference / benchmarking: replicas only -- no data-path collective (`shard_units`).
  * training: ONE exchange per step = gradient averaging.  `GradientBuckets` packs gradients into
    flat buckets as they become ready during backward and launches an asynchronous all-reduce per
    full bucket, so communication overlaps the rest of backward.  On MI355X the 7 xGMI links per GPU
    are point-to-point (~153 GB/s each): a ring all-reduce is bound by one link, so buckets are kept
    large (default 32 MiB: ~0.4 ms on the wire at 8 GPUs) to amortise launch latency, and small
    models (the aggregation alone is 4.2 MB of gradients) go out as a single bucket.
  * BatchNorm statistics in train mode: `sync_batchnorm(module)` converts to torch's SyncBatchNorm
    (per-layer all_gather of mean/invstd/count).
Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn


def init_distributed(backend=None):
    """Initialise from the torchrun environment.  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    return rank, world, local


def shard_units(n_units, rank, world, drop_last=False):
    """Indices of the stereo pairs / sequences this rank owns (contiguous-strided like
    DistributedSampler without shuffling: unit i goes to rank i % world)."""
    n = (n_units // world) * world if drop_last else n_units
    return list(range(rank, n, world))


class _SyncBatchNormFn(torch.autograd.Function):
    """Train-mode BatchNorm whose statistics span every rank of the group.  Forward: ONE all_gather of the per-rank
    [mean | biased var | count] (2C+1 floats), combined with the parallel-variance formula; backward: ONE all_reduce of
    [sum dy | sum dy*(x-mean)] (2C floats).  Weight / bias gradients stay local: the gradient exchange averages them with the
    rest (what torch's SyncBatchNorm and detectron2's NaiveSyncBatchNorm do)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        n_local = x.numel() // C
        var_l, mean_l = torch.var_mean(x, dims, unbiased=False)
        world = dist.get_world_size(group)
        pack = torch.cat([mean_l, var_l, x.new_full((1,), float(n_local))])
        allp = [torch.empty_like(pack) for _ in range(world)]
        dist.all_gather(allp, pack, group=group)
        allp = torch.stack(allp)                                     # [world, 2C+1]
        cnt = allp[:, -1:]                                           # [world, 1]
        n = cnt.sum()
        mean = (allp[:, :C] * cnt).sum(0) / n
        var = ((allp[:, C:2 * C] + (allp[:, :C] - mean) ** 2) * cnt).sum(0) / n
        invstd = torch.rsqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        xhat = (x - mean.view(shape)) * invstd.view(shape)
        ctx.save_for_backward(xhat, weight, invstd)
        ctx.group, ctx.n, ctx.dims, ctx.shape = group, n, dims, shape
        ctx.mark_non_differentiable(mean, var, n)
        y = xhat * weight.view(shape) + bias.view(shape) if weight is not None else xhat
        return y, mean, var, n

    @staticmethod
    def backward(ctx, gy, _gm, _gv, _gn):
        xhat, weight, invstd = ctx.saved_tensors
        dims, shape = ctx.dims, ctx.shape
        gw = (gy * xhat).sum(dims)
        gb = gy.sum(dims)
        pack = torch.cat([gb, gw])
        dist.all_reduce(pack, op=dist.ReduceOp.SUM, group=ctx.group)
        C = gb.numel()
        sum_dy, sum_dy_xhat = pack[:C] / ctx.n, pack[C:] / ctx.n
        w = weight if weight is not None else torch.ones_like(invstd)
        gx = (gy - sum_dy.view(shape) - xhat * sum_dy_xhat.view(shape)) * (invstd * w).view(shape)
        return gx, (gw if weight is not None else None), (gb if weight is not None else None), None, None


class SyncBatchNorm(nn.modules.batchnorm._BatchNorm):
    """BatchNorm{2,3}d with cross-rank batch statistics over `torch.distributed` (RCCL on the GPUs, gloo in the CPU tests) --
    the role of `sync_batchnorm=True` in the reference's trainer (projects/TemporalStereo/dist_train.py:94).  Same parameters,
    buffers and state-dict names as nn.BatchNorm*d; eval mode (and a single process) is plain batch_norm."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, process_group=None):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.process_group = process_group

    def _check_input_dim(self, input):
        if input.dim() < 2:
            raise ValueError("expected at least 2D input (got {}D input)".format(input.dim()))

    def forward(self, x):
        sync = self.training and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1
        if not sync:
            return super().forward(x)
        y, mean, var, n = _SyncBatchNormFn.apply(x, self.weight, self.bias, self.eps, self.process_group)
        if self.track_running_stats:
            with torch.no_grad():
                self.num_batches_tracked += 1
                m = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                self.running_mean.mul_(1 - m).add_(mean, alpha=m)
                self.running_var.mul_(1 - m).add_(var * (n / (n - 1)), alpha=m)
        return y


def sync_batchnorm(module, process_group=None):
    """Replace every nn.BatchNorm{1,2,3}d of `module` (in place, names and tensors kept) by SyncBatchNorm."""
    out = module
    if isinstance(module, nn.modules.batchnorm._BatchNorm) and not isinstance(module, SyncBatchNorm):
        out = SyncBatchNorm(module.num_features, module.eps, module.momentum, module.affine, module.track_running_stats, process_group)
        if module.affine:
            out.weight, out.bias = module.weight, module.bias
        out.running_mean, out.running_var, out.num_batches_tracked = module.running_mean, module.running_var, module.num_batches_tracked
        out.training = module.training
    for name, child in list(module.named_children()):
        new = sync_batchnorm(child, process_group)
        if new is not child:
            setattr(out, name, new)
    return out


class GradientBuckets:
    """Bucketed, backward-overlapped gradient averaging.

        gb = GradientBuckets(model.parameters())
        loss.backward()          # hooks fire, full buckets are all-reduced asynchronously
        gb.finish()              # wait, average, scatter back into .grad

    One backward per finish (gradient accumulation: call backward under `gb.paused()` for all but the last
    micro-batch).  Parameters that are not part of the autograd graph (FineAggregation.phi is declared but never
    used, fine.py:34) would keep their bucket from ever filling, so the first step finds them -- every rank
    agrees on the set through one MAX all-reduce of the "fired" flags -- and the buckets are rebuilt without
    them; their .grad stays None, as it does in the reference (Lightning's DDP runs with
    find_unused_parameters)."""

    def __init__(self, params, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.params = [p for p in params if p.requires_grad]
        self._unused_known = False
        self._paused = False
        self.launched_in_backward = 0         # buckets whose all-reduce went out before finish() (overlap evidence)
        self._layout(self.params)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _layout(self, params):
        # reverse registration order ~ the order gradients become ready in backward
        order = list(reversed(params))
        self.buckets, cur, cur_bytes = [], [], 0
        for p in order:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(cur)
        self._where, self._flat, self._ready = {}, [], []
        for bi, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            self._flat.append(torch.zeros(n, dtype=bucket[0].dtype, device=bucket[0].device))
            off = 0
            for p in bucket:
                self._where[id(p)] = (bi, off)
                off += p.numel()
            self._ready.append(0)
        self._handles = [None] * len(self.buckets)
        self._fired = set()
        self._accumulated = set()             # parameters that received gradient under paused() since the last finish()
        self._next = 0                        # buckets are all-reduced strictly in index order (the same order on every rank)

    def paused(self):
        """Context manager: backward passes inside accumulate into .grad without communication."""
        gb = self

        class _P:
            def __enter__(self_):
                gb._paused = True

            def __exit__(self_, *exc):
                gb._paused = False
                return False
        return _P()

    def _on_grad(self, p):
        if self.world == 1:
            return
        if self._paused:
            self._accumulated.add(id(p))      # its .grad holds a contribution that finish() must not drop if it does not fire again
            return
        if id(p) in self._fired:
            raise RuntimeError("GradientBuckets: a second backward() before finish() -- accumulate under gb.paused()")
        self._fired.add(id(p))
        if id(p) not in self._where:          # declared unused on the first step, used now
            raise RuntimeError("GradientBuckets: a parameter that received no gradient on the first step received one now")
        bi, off = self._where[id(p)]
        self._flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
        self._ready[bi] += 1
        # Launch order must not depend on the rank: a data-dependent branch can make a parameter fire on one rank and not on another,
        # and collectives issued in different orders on different ranks hang (or sum the wrong buffers).  So buckets go out strictly
        # by index; a full bucket behind one that is not full yet waits for it (or for finish()).
        # ... and not at all on the first step: the ranks first agree on the participating parameters (finish(): one all-reduce of the
        # "fired" flags), and that collective must be the first one on every rank whatever fired where.
        while self._unused_known and self._next < len(self.buckets) and self._ready[self._next] == len(self.buckets[self._next]):
            nb = self._next
            self._handles[nb] = dist.all_reduce(self._flat[nb], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launched_in_backward += 1
            self._next += 1

    def finish(self):
        """Wait for every bucket, write the averaged gradients back."""
        if self.world == 1:
            return
        first = not self._unused_known
        if first:
            # which parameters take part: agreed across ranks (a parameter used on any rank is reduced on all)
            flags = torch.tensor([1.0 if (id(p) in self._fired or id(p) in self._accumulated) else 0.0 for p in self.params], device=self.params[0].device)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
            used = [p for p, f in zip(self.params, flags.tolist()) if f > 0]
            used_ids = {id(p) for p in used}
        for bi, bucket in enumerate(self.buckets):            # the rest, in index order again
            if self._handles[bi] is None:
                for p in bucket:
                    if id(p) not in self._fired:              # no gradient on this rank in the last backward: contributes zero (a .grad
                        _, off = self._where[id(p)]           # left over from an earlier step is stale and must not be averaged in) --
                        if id(p) in self._accumulated and p.grad is not None:      # unless a paused() pass of THIS step accumulated into it
                            self._flat[bi][off:off + p.numel()].copy_(p.grad.reshape(-1))
                        else:
                            self._flat[bi][off:off + p.numel()].zero_()
                self._handles[bi] = dist.all_reduce(self._flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for bi, bucket in enumerate(self.buckets):
            self._handles[bi].wait()
            self._flat[bi].div_(self.world)
            for p in bucket:
                if first and id(p) not in used_ids:
                    continue                                      # outside the graph on every rank: .grad stays None
                _, off = self._where[id(p)]
                g = self._flat[bi][off:off + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            self._handles[bi] = None
            self._ready[bi] = 0
        self._fired = set()
        self._accumulated = set()
        self._next = 0
        if first:
            self._unused_known = True
            if len(used) != len(self.params):
                self._layout(used)

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_parameters(module, src=0, group=None):
    """Same initial weights on every rank (what DDP does at construction)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
