"""One data-parallel training step of the aggregation path, the counterpart of the reference's
`TemporalStereo.training_step` + `multi_frame_forward` (projects/TemporalStereo/TemporalStereo.py:130-168, :250-280) under
`pl.Trainer(strategy='ddp', sync_batchnorm=True, gradient_clip_val=0.1)` (dist_train.py:82-96):

    for every previous frame (FRAME_IDXS < 0; PREVIOUS_WITH_GRADIENT False):   eval() + no_grad forward      :268-274
        update_map: the temporal state moves into the next frame                                               :326-461
    current frame: train() forward with the carried state                                                      :276-278
    loss = smooth-L1 of the four disparities (rescaled to full size) + Wasserstein loss of the three levels    :139-150
    backward -> gradient averaging over the ranks -> clip 0.1 -> RMSprop step                                   sceneflow.yaml:21-24

The aggregation's inputs are the backbone's feature pyramids (out of scope: features are given, and `requires_grad` so that the
cost volume's backward towards them runs as it would under the backbone).  Everything on the data path is a HIP kernel behind an
autograd Function: cost volume, every conv -> BatchNorm -> activation wrapper (train-mode statistics from the BatchNorm kernels,
exchanged across ranks by dist.SyncBatchNorm; the eval frames with the BatchNorm folded into the convolution), transposed 2-D
convolutions, resize / pooling / sort + gather, candidates, offset head, upsamplers, both losses, clip + RMSprop.  What the framework
still runs per step is ~90 small launches (torch.cat of volumes, fills, bias reductions, copies).

Issued op by op the step is host-bound (~1,160 launches through the framework's autograd per T=2 step: 19 ms against ~12 ms of
device time).  `graph=True` captures previous frames + state update + forward + losses + backward ONCE into a
hipGraph and replays it per step on static copies of the inputs (the step copies each call's tensors into them); gradient
exchange, clipping and the optimizer stay outside.  ROCm 7.2's replay of pre-built AQL packets
("graph packet capture") computes garbage gradients from the second or third replay of a graph of this size on (loss finite,
gradients 1e35 / NaN; bit-stable with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0), so graph mode insists that this setting was in place
BEFORE the HIP runtime started: either in the environment the process was started with, or through `enable_graph_replay()`
called before anything touched the GPU.  Importing the package changes nothing in the environment.
"""
import contextlib
import os
import time

import torch

from . import dist as tsd
from . import functional as TF
from . import temporal
from .losses import DispSmoothL1Loss, WarssersteinDistanceLoss


_GRAPH_VAR = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_set_before_runtime = False


def _initial_environment():
    """The environment this process was exec'd with (what the HIP runtime's static initialisers can have seen at the earliest)."""
    try:
        with open("/proc/self/environ", "rb") as fh:
            items = fh.read().split(b"\0")
        return dict(kv.decode(errors="replace").split("=", 1) for kv in items if b"=" in kv)
    except OSError:
        return {}


def graph_replay_safe():
    """True when DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is known to have been set before the HIP runtime read its settings."""
    if os.environ.get(_GRAPH_VAR) != "0":
        return False
    return _set_before_runtime or _initial_environment().get(_GRAPH_VAR) == "0"


def enable_graph_replay():
    """Explicit opt-in for TrainStep(graph=True): put DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 into the environment.  Only effective
    before the HIP runtime starts, so it raises once the GPU has been touched (unless the setting was there from the start)."""
    global _set_before_runtime
    if graph_replay_safe():
        return
    if torch.cuda.is_initialized():
        raise RuntimeError("enable_graph_replay() must run before the first GPU call of the process: the HIP runtime has already "
                           "read its settings (start the process with %s=0 instead)" % _GRAPH_VAR)
    os.environ[_GRAPH_VAR] = "0"
    _set_before_runtime = True


class ClipRMSprop:
    """clip_grad_norm_(max_norm) + torch.optim.RMSprop(lr, alpha, eps) (no momentum, not centred, no weight decay: the reference's
    optimizer, sceneflow.yaml:21-24, under gradient_clip_val=0.1, dist_train.py:94) as TWO launches over a pointer table of all
    parameters (ts_clip_rmsprop_step) instead of ~30 multi-tensor launches and their host-side bookkeeping.  Parameters without a
    gradient are skipped, as the framework's optimizer does.  Deterministic (fixed-order norm reduction)."""

    def __init__(self, params, lr=1e-3, alpha=0.99, eps=1e-8, max_norm=0.1):
        import numpy as np
        self.params = list(params)
        self.lr, self.alpha, self.eps, self.max_norm = float(lr), float(alpha), float(eps), float(max_norm or 0.0)
        self.param_groups = [dict(lr=self.lr, params=self.params)]          # the part of the optimizer interface schedulers use
        self.square_avg = [torch.zeros_like(p, memory_format=torch.preserve_format) for p in self.params]
        n = len(self.params)
        self._dtype = np.dtype([("param", "<u8"), ("grad", "<u8"), ("sq", "<u8"), ("n", "<i8")])
        self._host = [torch.empty(n * 32, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else torch.empty(n * 32, dtype=torch.uint8)
                      for _ in range(2)]
        self._turn = 0
        self._copied = [None, None]          # event behind the last non-blocking upload from each pinned table
        self._table = self._ws = None
        self._static_grads = None
        self.steps = 0

    def state_dict(self):
        """square_avg per parameter (in parameter order), hyper-parameters and the step count: what a checkpoint needs to resume
        (the reference's Lightning checkpoints carry the optimizer state)."""
        return dict(square_avg=[t.detach().clone() for t in self.square_avg], lr=float(self.param_groups[0]["lr"]), alpha=self.alpha,
                    eps=self.eps, max_norm=self.max_norm, steps=self.steps)

    def load_state_dict(self, state):
        sq = state["square_avg"]
        if len(sq) != len(self.square_avg) or any(a.shape != b.shape for a, b in zip(sq, self.square_avg)):
            raise ValueError("ClipRMSprop.load_state_dict: state does not match the parameters")
        with torch.no_grad():
            for dst, src in zip(self.square_avg, sq):
                dst.copy_(src)
        self.param_groups[0]["lr"] = self.lr = float(state.get("lr", self.lr))
        self.alpha, self.eps = float(state.get("alpha", self.alpha)), float(state.get("eps", self.eps))
        self.max_norm, self.steps = float(state.get("max_norm", self.max_norm)), int(state.get("steps", 0))

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def total_norm(self):
        """Gradient norm of the last step (before clipping), a 0-d device tensor."""
        if self._ws is None:
            return None
        off = len(self.params) * 8 * 4                       # behind the 8 partial sums per tensor (ts_clip_rmsprop_step)
        return self._ws[off:off + 4].view(torch.float32)[0]

    def step(self):
        from . import _lib
        from .functional import _stream
        ps = self.params
        if not ps:
            return
        dev = ps[0].device
        n = len(ps)
        L = _lib.lib()
        if self._table is None:
            self._table = torch.empty(n * 32, dtype=torch.uint8, device=dev)
            nb = int(L.ts_clip_rmsprop_workspace_bytes(n))
            self._ws = torch.empty((nb + 3) // 4 * 4, dtype=torch.uint8, device=dev)
        grads = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in ps)
        if grads != self._static_grads:                      # eager steps: fresh gradient tensors every backward; graph replays: static
            host = self._host[self._turn]
            if self._copied[self._turn] is not None:
                self._copied[self._turn].synchronize()     # the upload that last read this pinned table has completed
            rec = host.numpy().view(self._dtype)
            for i, p in enumerate(ps):
                g = p.grad
                if g is not None and (not g.is_contiguous() or g.dtype != torch.float32):
                    raise RuntimeError("ClipRMSprop: gradients must be dense fp32")
                rec[i] = (p.data_ptr(), grads[i], self.square_avg[i].data_ptr(), p.numel())
            self._table.copy_(host, non_blocking=True)
            if dev.type == "cuda":
                ev = torch.cuda.Event()
                ev.record()
                self._copied[self._turn] = ev
            self._turn ^= 1
            self._static_grads = grads
        lr = float(self.param_groups[0]["lr"])
        _lib.check(L.ts_clip_rmsprop_step(_lib.ptr(self._table), n, self.max_norm, lr, self.alpha, self.eps, _lib.ptr(self._ws),
                                          self._ws.numel(), _stream()), "ts_clip_rmsprop_step")
        # the kernel writes through raw pointers: tell autograd (and everything that stamps (storage, version), e.g. the
        # InferenceEngine's stale-weights check) that the parameters changed, as torch.optim.RMSprop's in-place ops would.
        # Note: p.grad is left UNclipped (clip_grad_norm_ scales it in place); the clipped step is applied inside the kernel.
        torch.autograd.graph.increment_version(ps)
        self.steps += 1


# the reference's training configuration (projects/TemporalStereo/configs/sceneflow.yaml:21-24, :61-67)
REFERENCE_L1_WEIGHTS = (2.0, 1.0, 0.7, 0.5)
REFERENCE_WARS_WEIGHTS = (1.0, 0.7, 0.5)
REFERENCE_WARS_GLOBAL_WEIGHT = 2.0
REFERENCE_LR = 1e-3


class TrainStep:
    def __init__(self, net, max_disp=192, local_map_size=1, lr=REFERENCE_LR, clip=0.1, sync_bn=True, bucket_bytes=32 << 20, baseline=1.0,
                 graph=False, l1_weights=REFERENCE_L1_WEIGHTS, l1_global_weight=1.0, wars_weights=REFERENCE_WARS_WEIGHTS,
                 wars_global_weight=REFERENCE_WARS_GLOBAL_WEIGHT, fused_optimizer=True):
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.graph = bool(graph)
        if self.graph and not graph_replay_safe():
            raise RuntimeError("TrainStep(graph=True) needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in place before the HIP runtime starts "
                               "(ROCm 7.2 replays pre-built graph packets incorrectly; see train.py): start the process with it, or "
                               "call temporalstereo_amd.train.enable_graph_replay() before the first GPU call")
        # sync_bn: True -> the statistics exchanges are torch.distributed collectives (RCCL / gloo), one per layer and direction;
        # "peer" -> kernels over the peer-mapped mailboxes of the node's ranks (peer.py, csrc/peer.hip): no communicator launch per
        # layer, and -- being kernels -- part of a captured graph, which makes the replayed step legal for world > 1 (round 4; rounds
        # 2-3 refused graph=True with SyncBatchNorm).
        if self.graph and self.world > 1 and sync_bn and sync_bn != "peer":
            # collectives are not captured: a replayed step could only normalise with per-rank statistics, which is NOT what the
            # reference trains with (sync_batchnorm=True, dist_train.py:94) -- refuse rather than change semantics silently
            raise RuntimeError("TrainStep(graph=True) across %d ranks needs sync_bn='peer' (statistics exchanged by kernels, which a hipGraph "
                               "can replay; torch.distributed collectives cannot be), or sync_bn=False to accept per-rank BatchNorm "
                               "explicitly" % self.world)
        self.sync_bn = bool(sync_bn) and self.world > 1
        self.peer = None
        from . import peer as _peer
        stale = _peer.installed()
        if stale is not None and (getattr(stale, "closed", False) or getattr(stale, "owner", lambda: None)() is None):
            # a PeerGroup left installed by a TrainStep that is gone (or closed) is not this step's: its exchanges would go through
            # dead mailboxes.  The group of a LIVE step (a validation step built beside a training one) stays installed.
            _peer.install(None)
        if self.sync_bn:
            net = tsd.sync_batchnorm(net)
            if sync_bn == "peer":
                self.peer = _peer.PeerGroup()
                import weakref
                self.peer.owner = weakref.ref(self)
                _peer.install(self.peer)
        self.net = net
        tsd.broadcast_parameters(net)
        self.l1 = DispSmoothL1Loss(max_disp=max_disp, rescale=True, global_weight=l1_global_weight,
                                   weights=list(l1_weights) if l1_weights is not None else None)
        self.wars = WarssersteinDistanceLoss(max_disp=max_disp, global_weight=wars_global_weight,
                                             weights=list(wars_weights) if wars_weights is not None else None)
        self.local_map_size, self.clip, self.baseline = local_map_size, clip, baseline
        self.params = [p for p in net.parameters() if p.requires_grad]
        on_gpu = bool(self.params) and self.params[0].is_cuda
        self.fused_optimizer = bool(fused_optimizer) and on_gpu
        self.opt = ClipRMSprop(self.params, lr=lr, max_norm=clip) if self.fused_optimizer else torch.optim.RMSprop(self.params, lr=lr)
        self.buckets = tsd.GradientBuckets(self.params, bucket_bytes=bucket_bytes) if self.world > 1 and not self.graph else None
        self._g = self._static = self._loss = None
        self.timings = {}
        self._modules = list(net.modules())
        self._stat_buffers = [b for b in net.buffers() if b.is_cuda]
        # Kernel layouts of all convolution weights in one launch per step instead of ~280: 5 ms of host time in the eager step
        # (28 -> 22.7 ms).  NOT in the replayed step, measured again in round 3 (ms per replay): a layout launch in front of every
        # convolution call 13.4; one launch per step 14.0-14.3; one launch per 16 / 32 / 64 consecutive requests, issued just ahead of
        # them (so that the layouts are still in the Infinity Cache when read) 14.2 / 14.1 / 14.1.  The ~270 small launches are not what
        # the replay waits for, and with kept layouts the convolution kernels themselves run ~7 % slower (rocprof: 4.42 vs 4.12 ms per
        # step) -- the per-call buffers are one recycled block of the graph's pool, hot in every cache level.  TS_TRAIN_GRAPH_LAYOUTS=1
        # selects the one-launch form for the A/B (with TS_SPLIT_FIRST_LAYER=0: the split first layer's weight halves are new tensors
        # every step, which the kept table cannot register inside a capture).
        self.layouts = TF.WeightLayouts() if (not self.graph or os.environ.get("TS_TRAIN_GRAPH_LAYOUTS", "0") != "0") else None
        # the previous frames run in eval() / no_grad: their conv -> BatchNorm -> activation wrappers become one convolution launch
        # each, the BatchNorm folded into its epilogue; all folds of the step are recomputed by one launch (functional.BNFolds)
        self.folds = TF.BNFolds() if (on_gpu and os.environ.get("TS_TRAIN_BN_FOLDS", "1") != "0") else None
        # Round 5: the previous frames through the INFERENCE form of the same network (aggregation.native: bf16-split convolutions,
        # the warp-commuted first layers, channel-sliced outputs instead of torch.cat, merged heads -- ~110 launches instead of ~210
        # per frame), its folded arrays re-made in place from the current parameters / running statistics once per step by native.Tape
        # (three table-driven launches + one per bf16-split copy; round 4 tried this with a full rebuild per step -- ~2,000
        # framework launches -- and lost).  Built lazily at the first step with previous frames; TS_TRAIN_NATIVE_PREV=0 keeps the
        # module path for them.
        # one wgrad_finish launch per step instead of one per layer (functional.WgradDefer); not when gradient buckets go out during
        # backward (their hooks read .grad as soon as it arrives)
        self.wgrad_defer = TF.WgradDefer() if (on_gpu and self.buckets is None and os.environ.get("TS_TRAIN_WGRAD_DEFER", "1") != "0") else None
        self._native_prev = None
        self._use_native_prev = on_gpu and os.environ.get("TS_TRAIN_NATIVE_PREV", "1") != "0"
        self._native_prev_ran = False
        # (Tried and dropped in round 3: the weight-gradient launches on a forked side stream inside the capture -- they depend only on
        # dy, 15 % of the step's device time, small grids.  The replayed graph got SLOWER, 15.0 vs 13.4 ms: forked captures replay
        # badly on ROCm 7.2, as the inference graph already showed)

    def close(self):
        """Collective when sync_bn='peer' (a barrier: nobody unmaps a mailbox a peer may still write): uninstalls and releases the
        peer group.  Idempotent; the step object must not be called afterwards."""
        pg, self.peer = self.peer, None
        if pg is not None:
            from . import peer as _peer
            if _peer.installed() is pg:
                _peer.install(None)
            pg.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _previous_frame_network(self):
        """The inference form of the network for the eval / no_grad frames, its folded arrays brought up to date with the current
        parameters and running statistics (in place; inside a capture this is part of the captured step).  None: module path."""
        if not self._use_native_prev:
            return None
        if self._native_prev is None:
            from .aggregation.engine import InferenceEngine
            from .aggregation.native import NativeAggregator
            try:
                self._set_training(False)
                if self.graph:
                    # captured step: the aggregator itself, on one stream (forked captures replay badly on ROCm 7.2)
                    agg = NativeAggregator(self.net)
                    agg.overlap = False
                    eng = None
                else:
                    # eager step: the pass replayed from a recorded launch plan (one host call instead of ~110 issued from Python)
                    eng = InferenceEngine(self.net, backend="native", replay="plan", inputs="copy")
                    agg = eng.net
                if agg.tape.unsupported:
                    raise NotImplementedError("; ".join(agg.tape.unsupported))
                if agg.tape._tables is None and not agg.tape.unsupported:
                    # the re-fold tables go up NOW (a numpy table + a pageable upload): never inside a capture, whatever the warm-up count
                    agg.tape._upload(next(self.net.parameters()).device)
                self._native_prev = (agg, eng)
                return eng if eng is not None else agg   # just built from the current values
            except NotImplementedError:                  # a model the inference form does not cover: the module path serves it
                self._use_native_prev = False
                return None
        agg, eng = self._native_prev
        if eng is not None:
            eng.refold_in_stream()
            return eng
        agg.refresh_weights()
        return agg

    def _set_training(self, mode):
        """net.train(mode) without nn.Module.__setattr__'s bookkeeping on ~640 modules (2.4 ms of host time per step)."""
        for m in self._modules:
            m.__dict__["training"] = mode

    # ------------------------------------------------------------------------------------------------------------
    def _forward_backward(self, frames, gt, K, poses):
        """frames: list of (left_feats, right_feats, left_image, right_image), oldest first; poses[t] = (T_now, inv_T_past)."""
        net = self.net
        with (self.layouts if self.layouts is not None else contextlib.nullcontext()), \
                (self.wgrad_defer if self.wgrad_defer is not None else contextlib.nullcontext()), \
                (self.folds if (self.folds is not None and len(frames) > 1) else contextlib.nullcontext()):
            if self.layouts is not None:
                self.layouts.refresh()
            if self.folds is not None and len(frames) > 1:
                self.folds.refresh()
            info = {}
            prev_net = self._previous_frame_network() if len(frames) > 1 else None
            for t, fr in enumerate(frames[:-1]):
                self._set_training(False)
                with torch.no_grad():
                    if prev_net is not None and not self._native_prev_ran:
                        # the first CALL can still refuse (a geometry or stride the inference form rejects): the module path serves it
                        try:
                            out = prev_net(fr[0], fr[1], fr[2], fr[3], dict(info))
                            self._native_prev_ran = True
                        except (NotImplementedError, RuntimeError, ValueError):
                            if torch.cuda.is_current_stream_capturing():
                                raise
                            self._use_native_prev, self._native_prev, prev_net = False, None, None
                            out = net(fr[0], fr[1], fr[2], fr[3], dict(info))
                        info = out[5]
                    else:
                        info = (prev_net if prev_net is not None else net)(fr[0], fr[1], fr[2], fr[3], dict(info))[5]
                    H, W = fr[2].shape[-2:]
                    info = temporal.update_map(dict(info), K, poses[t + 1][0], poses[t + 1][1], self.baseline, H, W,
                                               use_past_cost=True, local_map_size=self.local_map_size)
            self._set_training(True)
            cur = frames[-1]
            state = {k: v for k, v in info.items() if k in ("cost_memory", "use_past_cost", "local_map", "local_map_size") and v is not None}
            disps, costs, samples, offs, _, _ = net(cur[0], cur[1], cur[2], cur[3], state)
            if disps[0].is_cuda:
                # the loss objects' terms (losses.py: same kernels, same weights) combined by ONE weighted sum instead of a
                # multiplication per term each way, a stack and a sum: weights[i] * global_weight multiplied on the host
                l1w = self.l1.weights if self.l1.weights is not None else [1.0] * len(disps)
                ww = self.wars.weights if self.wars.weights is not None else [1.0] * len(costs)
                terms = [self.l1.loss_per_level(d, gt) for d in disps] + \
                        [self.wars.loss_per_level(c, o, s, gt) for c, o, s in zip(costs, offs, samples)]
                weights = [float(w) * float(self.l1.global_weight) for w in l1w[:len(disps)]] + \
                          [float(w) * float(self.wars.global_weight) for w in ww[:len(costs)]]
                total = TF.weighted_total(terms, weights)
            else:
                losses = {}
                losses.update(self.l1(disps, gt))
                losses.update(self.wars(costs, offs, samples, gt))
                total = torch.stack(list(losses.values())).sum()
            total.backward()
            if self.wgrad_defer is not None:
                self.wgrad_defer.flush()                 # every deferred weight gradient is final from here on
        return total.detach()

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _tensors(tree):
        if torch.is_tensor(tree):
            yield tree
        elif isinstance(tree, (list, tuple)):
            for t in tree:
                yield from TrainStep._tensors(t)
        elif tree is not None and not isinstance(tree, (int, float, bool, str)):
            # a numpy pose / a dict would be baked into the capture as a constant and silently ignored on later calls
            raise TypeError("TrainStep(graph=True): inputs must be tensors in lists / tuples, got %s" % type(tree).__name__)

    @staticmethod
    def _clone(tree):
        if torch.is_tensor(tree):
            return tree.detach().clone().requires_grad_(tree.requires_grad)
        if isinstance(tree, (list, tuple)):
            return type(tree)(TrainStep._clone(t) for t in tree)
        return tree

    def _capture(self, args):
        self._static = self._clone(args)
        # the warm-up passes must leave no trace: BatchNorm's running statistics are put back afterwards
        buffers = [(b, b.detach().clone()) for b in self.net.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                                      # warm-up: allocator, solver choices, lazy initialisations
                self.opt.zero_grad(set_to_none=True)
                self._forward_backward(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        for b, saved in buffers:
            b.copy_(saved)
        torch.cuda.synchronize()
        self.opt.zero_grad(set_to_none=True)
        for t in self._tensors(self._static):
            t.grad = None
        self._g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g):
            self._loss = self._forward_backward(*self._static)

    def bound_inputs(self):
        """graph=True, after the first call: the (frames, gt, K, poses) tensors the captured step reads.  A producer that writes the
        next sample INTO them (the data pipeline's device-side collate, the backbone's outputs) and passes them back in saves the
        ~25 device-to-device copies per step that arbitrary inputs need -- the `inputs='bind'` contract of InferenceEngine."""
        if not self.graph or self._static is None:
            raise RuntimeError("bound_inputs(): a graph=True step, after its first call")
        return self._static

    def _all_reduce_flat(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        flat = torch._utils._flatten_dense_tensors(grads)
        torch.distributed.all_reduce(flat)
        flat.div_(self.world)
        for g, r in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(r)

    # ------------------------------------------------------------------------------------------------------------
    def __call__(self, frames, gt, K, poses):
        """One optimisation step.  Returns the (local) loss as a 0-d device tensor."""
        t0 = time.perf_counter()
        if self.graph:
            args = (frames, gt, K, poses)
            if self._g is None:
                self._capture(args)
            else:
                dsts, srcs = list(self._tensors(self._static)), list(self._tensors(args))
                if len(dsts) != len(srcs) or any(d.shape != s.shape or d.dtype != s.dtype for d, s in zip(dsts, srcs)):
                    raise RuntimeError("TrainStep(graph=True): this call's inputs do not have the structure / shapes / dtypes the graph was "
                                       "captured with (%d tensors); build a new TrainStep for a new geometry" % len(dsts))
                for dst, src in zip(dsts, srcs):
                    # (a caller that fills `bound_inputs()` in place copies nothing; a VIEW of the bound buffer with other strides or
                    # offset -- flipped, transposed -- shares its data_ptr but not its contents and is copied)
                    if not (dst is src or (dst.data_ptr() == src.data_ptr() and dst.stride() == src.stride()
                                           and dst.storage_offset() == src.storage_offset())):
                        dst.detach().copy_(src, non_blocking=True)
            self._g.replay()
            loss = self._loss.clone()                                # the graph's own buffer is overwritten by the next replay
            t1 = time.perf_counter()
            if self.world > 1:
                self._all_reduce_flat()
        else:
            self.opt.zero_grad(set_to_none=True)
            loss = self._forward_backward(frames, gt, K, poses)
            t1 = time.perf_counter()
            if self.buckets is not None:
                self.buckets.finish()
        t2 = time.perf_counter()
        if self.fused_optimizer:
            self.opt.step()                                          # clip + RMSprop: two launches
        else:
            if self.clip:
                torch.nn.utils.clip_grad_norm_(self.params, self.clip)
            self.opt.step()
        if self._stat_buffers:
            # BatchNorm running statistics are updated inside the statistics kernels (raw pointers; under graph replay no host code
            # runs at all): bump their versions like the in-place framework update would, for everything that stamps (storage, version)
            torch.autograd.graph.increment_version(self._stat_buffers)
        if self.peer is not None:
            # a time-out of an exchange leaves garbage statistics behind and kills the group: noticed here one step late (no
            # synchronisation: the err word is copied behind the queued work and read at the next step), and training stops
            self.peer.poll()
        t3 = time.perf_counter()
        # host-side issue times (the device runs behind them); the exchange entry includes waiting for the reduced buckets
        self.timings = dict(forward_backward_issue_ms=(t1 - t0) * 1e3, exchange_ms=(t2 - t1) * 1e3, clip_step_issue_ms=(t3 - t2) * 1e3)
        return loss
