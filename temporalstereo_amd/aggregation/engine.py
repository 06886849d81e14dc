"""Inference engine of the aggregation hot path: one recorded launch plan (or hipGraph) per input signature.

The reference runs its forward eagerly (~300 framework ops per pair, SURVEY.md section 7 "tiny
problem sizes": launch gaps dominate at batch 1).  Here the whole coarse -> fine -> precise pass is
captured ONCE into a HIP graph (torch.cuda.CUDAGraph is hipGraph on ROCm; our C-ABI launches go to
the capturing stream like any other kernel) and replayed per frame: no Python, no allocator, no
per-op launch latency on the critical path.  BatchNorm is in eval mode, so the captured kernels are
the folded/fused inference variants.

Static-buffer contract: inputs are copied into graph-owned buffers, outputs are graph-owned
tensors that the next call overwrites (clone what must outlive the next frame).  The temporal state
`prev_info` is part of the signature: a graph is built per (shapes, which state entries exist).
"""
import os

import torch


def _sig_of(obj):
    if torch.is_tensor(obj):
        return ("T",) + tuple(obj.shape)
    if isinstance(obj, dict):
        return ("D",) + tuple((k, _sig_of(obj[k])) for k in sorted(obj, key=str))
    if isinstance(obj, (list, tuple)):
        return ("L",) + tuple(_sig_of(o) for o in obj)
    return ("V", obj)


def _ptrs_of(obj):
    if torch.is_tensor(obj):
        return (obj.data_ptr(),)
    if isinstance(obj, dict):
        return tuple(p for k in sorted(obj, key=str) for p in _ptrs_of(obj[k]))
    if isinstance(obj, (list, tuple)):
        return tuple(p for o in obj for p in _ptrs_of(o))
    return ()


def _clone_static(obj):
    if torch.is_tensor(obj):
        return obj.detach().clone().contiguous()
    if isinstance(obj, dict):
        return {k: _clone_static(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_clone_static(o) for o in obj)
    return obj


def _copy_into(dst, src):
    if torch.is_tensor(dst):
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _copy_into(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _copy_into(d, s)


class _Captured:
    def __init__(self, graph, static_in, static_out):
        self.graph, self.static_in, self.static_out = graph, static_in, static_out

    def replay(self):
        self.graph.replay()


class _Recorded:
    """A pass recorded into a native plan (csrc/plan.hip): same static-buffer contract as a graph."""

    def __init__(self, recorder, static_in, static_out):
        self.recorder, self.static_in, self.static_out = recorder, static_in, static_out

    def replay(self):
        self.recorder.run()


_next_slot = [0]


def _new_event_slot():
    """A process-wide event slot of the native library (ts_event_record / ts_event_wait)."""
    if _next_slot[0] >= 256:
        raise RuntimeError("out of event slots")
    _next_slot[0] += 1
    return _next_slot[0] - 1


class InferenceEngine:
    """`engine(left_feats, right_feats, left_image, right_image, prev_info)` -> same tuple as
    TEMPORALSTEREO.forward, executed as a hipGraph replay."""

    def __init__(self, net, warmup=3, backend="native", graph=None, replay=None, inputs="copy", private_streams=False,
                 pipeline=1):
        """backend 'native': every stage on libts_hip.so kernels (aggregation.native);
        backend 'module': the nn.Module forward (torch/MIOpen convolutions + HIP K1/K4).
        replay: 'plan'  -- record the pass once into a native launch plan and re-issue it with one host
                           call per frame (native backend only; two-stream overlap stays on) [default
                           for 'native'];
                'graph' -- capture into a hipGraph [default for 'module'];
                'eager' -- run the backend directly.
        graph=True/False is the older spelling of replay='graph'/'eager'.
        inputs: 'copy' -- every call copies its arguments into the replay's own static buffers;
                'bind' -- the replay is bound to the tensors of the first call: the producer (the
                          backbone) writes each frame's features into those same tensors, so nothing is
                          copied; a call with different storage records a new plan for it.
        private_streams: the native backend's two helper streams are shared by every engine of a device;
                True gives this engine its own pair (several passes in flight on one GPU).
        pipeline: N = 2, 3 or 4 keeps N passes in flight (replay='plan', inputs='bind' only).  The pass is recorded N times,
                on N sets of buffers used in turn, as a four-stage pipeline over the engine's streams (coarse level |
                fine level | wide UNet half | 1/4-level tail on the caller's stream); none of the engine's streams waits
                for the caller's stream, where the tail of the previous call is still running on other buffers:
                consecutive, independent frames overlap (single-frame mode; a temporal sequence has a true dependency
                from frame to frame and should use pipeline=1).  Contract: the bound input tensors are complete on
                the device when the call is made, and the outputs of a call stay valid for N-1 further calls."""
        if inputs not in ("copy", "bind"):
            raise ValueError("inputs must be 'copy' or 'bind'")
        self.bind = inputs == "bind"
        if replay is None:
            replay = ("graph" if graph else "eager") if graph is not None else ("plan" if backend == "native" else "graph")
        if replay not in ("plan", "graph", "eager") or (replay == "plan" and backend != "native"):
            raise ValueError("replay must be 'plan' (native backend), 'graph' or 'eager'")
        graph = replay == "graph"
        if any(p.device.type != "cuda" for p in net.parameters()):
            raise RuntimeError("InferenceEngine needs the module on the GPU (there is no CPU path)")
        net = net.eval()
        self.module = net
        self.device = next(net.parameters()).device
        if backend == "native":
            from .native import NativeAggregator
            self.net = NativeAggregator(net, private_streams=private_streams)
            # cross-stream edges inside a captured graph replay far slower than they run eagerly on
            # ROCm 7.2 (measured 4.7 ms vs 1.9 ms per pair): a captured pass stays on one stream
            self.net.overlap = not graph
        elif backend == "module":
            self.net = net
        else:
            raise ValueError("backend must be 'native' or 'module'")
        self.backend, self.use_graph, self.replay = backend, graph, replay
        self.warmup = warmup
        self._graphs = {}
        if pipeline not in (1, 2, 3, 4) or (pipeline > 1 and not (replay == "plan" and self.bind)):
            raise ValueError("pipeline>1 needs replay='plan' and inputs='bind'")
        self.pipeline = pipeline
        self._slot_of = {}          # one named event per recorded plan: "the pass that last used these buffers is done"
        self._turn = {}
        # The native backend works on FOLDED COPIES of the weights and every recorded plan bakes in pointers to them:
        # a load_state_dict / optimizer step / .to() after construction must re-fold (demo.py:250 of the reference loads
        # the checkpoint after building the model).  Exact trigger: a load_state_dict hook; cheap per-call check: the
        # (storage, version) stamp of a spread of sentinel tensors -- every in-place update of the whole model (optimizer
        # step, load_state_dict's copy_) bumps all versions, .to() moves all storages; an edit of a single tensor by
        # hand is what refresh() is for.
        self._tensors = list(net.parameters()) + list(net.buffers())
        self._built_ptrs = [t.data_ptr() for t in self._tensors]
        self._sentinels = self._tensors[::16] + self._tensors[-1:]
        self._stamp = self._weights_stamp()
        self._stale = False
        self._hook = net.register_load_state_dict_post_hook(self._on_load_state_dict)

    def _on_load_state_dict(self, module, incompatible_keys):
        self._stale = True

    def _weights_stamp(self, full=False):
        return tuple((t.data_ptr(), t._version) for t in (self._tensors if full else self._sentinels))

    def refresh(self):
        """Re-read the module's parameters and buffers: re-folds BatchNorm / re-lays the weights (native backend) and
        drops every recorded plan / captured graph (their baked-in pointers and constants are stale).  Called
        automatically when the weights are seen to have changed; call it yourself after editing single tensors."""
        if self.module.training:
            raise RuntimeError("InferenceEngine runs the eval-mode (folded BatchNorm) pass: call module.eval() first")
        dev = next(self.module.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("InferenceEngine needs the module on the GPU (there is no CPU path)")
        tensors = list(self.module.parameters()) + list(self.module.buffers())
        same_storage = dev == self.device and len(tensors) == len(self._tensors) and \
            all(a is b and a.data_ptr() == p for a, b, p in zip(tensors, self._tensors, self._built_ptrs))
        if self.backend == "native" and same_storage and not self.net.tape.unsupported and os.environ.get("TS_ENGINE_REFOLD_IN_PLACE", "1") != "0":
            # the same tensors with new values (optimizer step, load_state_dict): every folded array re-made IN PLACE by a few
            # table-driven launches (native.Tape); the recorded plans / captured graphs keep pointing at the right arrays, so
            # nothing is dropped or re-recorded
            torch.cuda.synchronize(self.device)    # passes in flight on the engine's own streams still read the arrays ...
            self.net.refresh_weights()
            torch.cuda.synchronize(self.device)    # ... and the next pass's helper streams do not wait for the caller's stream
            self._stamp = self._weights_stamp()
            self._stale = False
            return
        torch.cuda.synchronize(self.device)        # passes still in flight read the buffers about to be dropped
        self.device = dev
        if self.backend == "native":
            self.net.device = dev
            with torch.cuda.device(dev):
                self.net._build(self.module)
                torch.cuda.synchronize(dev)        # helper streams do not wait for the caller's stream, where the fold just ran
        self._graphs.clear()
        self._turn.clear()
        self._tensors = list(self.module.parameters()) + list(self.module.buffers())
        self._built_ptrs = [t.data_ptr() for t in self._tensors]
        self._sentinels = self._tensors[::16] + self._tensors[-1:]
        self._stamp = self._weights_stamp()
        self._stale = False

    def refold_in_stream(self):
        """In-place re-fold (native.Tape) on the CALLER's stream without host synchronisation, for a caller that issues its passes
        one at a time on that same stream (pipeline == 1: every pass forks its helper streams from the caller's stream and joins
        them back, so the re-fold is ordered between two passes) -- train.TrainStep does this once per step for the previous
        frames.  The recorded plans stay valid."""
        if self.backend != "native" or self.pipeline != 1:
            raise RuntimeError("refold_in_stream: native backend, one pass at a time")
        if self._stale or self.net.tape.unsupported or any(t.data_ptr() != p for t, p in zip(self._tensors, self._built_ptrs)):
            was = self.module.training
            self.module.train(False)
            try:
                self.refresh()
            finally:
                self.module.train(was)
            return
        self.net.refresh_weights()
        self._stamp = self._weights_stamp()

    def _capture(self, args):
        static_in = args if self.bind else _clone_static(args)
        # eager warm-up on a side stream: MIOpen solver selection, lazy inits, allocator growth
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):
                self.net(static_in[0], static_in[1], static_in[2], static_in[3], dict(static_in[4]))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            out = self.net(static_in[0], static_in[1], static_in[2], static_in[3], dict(static_in[4]))
        return _Captured(graph, static_in, out)

    def _record(self, args):
        from .. import _lib
        static_in = args if self.bind else _clone_static(args)
        with torch.no_grad():
            for _ in range(max(self.warmup - 1, 0)):       # allocator growth, lazy initialisation
                self.net(static_in[0], static_in[1], static_in[2], static_in[3], dict(static_in[4]))
            torch.cuda.synchronize()
            rec = _lib.Recorder()
            with rec:
                out = self.net(static_in[0], static_in[1], static_in[2], static_in[3], dict(static_in[4]))
        torch.cuda.synchronize()
        return _Recorded(rec, static_in, out)

    def __call__(self, left_feats, right_feats, left_image, right_image, prev_info):
        if self._stale or self._weights_stamp() != self._stamp:
            self.refresh()
        with torch.cuda.device(self.device):
            return self._run(left_feats, right_feats, left_image, right_image, prev_info)

    # ------------------------------------------------------------------------------------------------ two-phase calls
    def begin(self, left_feats, right_feats, left_image, right_image):
        """First half of a pass: everything that depends on the frame's features / images only (see NativeAggregator.begin).
        For temporal sequences: call begin(frame t+1) right after finish(frame t) was issued, then update_map, then
        finish(handle, state) -- on the device the early part of t+1 overlaps the tail and the state update of t.
        Needs backend='native', replay='plan', inputs='bind' (the bound tensors must be complete on the device when begin is
        called, and stay untouched until finish has been issued)."""
        if not (self.backend == "native" and self.replay == "plan" and self.bind):
            raise RuntimeError("begin/finish need backend='native', replay='plan', inputs='bind'")
        if self._stale or self._weights_stamp() != self._stamp:
            self.refresh()
        with torch.cuda.device(self.device):
            args = (list(left_feats), list(right_feats), left_image, right_image)
            key = _sig_of(args) + (_ptrs_of(args),) + (torch.cuda.current_stream().cuda_stream,)
            turn = self._turn.get(key, 0)
            self._turn[key] = (turn + 1) % 2                      # two buffer sets, used alternately
            key = key + ("begin", turn)
            cap = self._graphs.get(key)
            if cap is None:
                from .. import _lib
                slots = (_new_event_slot(), _new_event_slot())
                with torch.no_grad():
                    self.net.begin(*args, slots)                    # allocator growth, lazy initialisation
                    torch.cuda.synchronize()
                    rec = _lib.Recorder()
                    with rec:
                        ctx = self.net.begin(*args, slots)
                torch.cuda.synchronize()
                cap = self._graphs[key] = _Recorded(rec, args, ctx)
                cap.late = {}
            else:
                cap.replay()
            return cap

    def finish(self, handle, prev_info):
        """Second half: candidate merges, fine level, 1/4-level tail.  Returns what __call__ returns."""
        with torch.cuda.device(self.device):
            state = {k: v for k, v in prev_info.items()
                     if k in ("cost_memory", "use_past_cost", "local_map", "local_map_size") and v is not None}
            sig = _sig_of(state)
            late = handle.late.get(sig)
            if late is None:
                from .. import _lib
                static_state = _clone_static(state)
                with torch.no_grad():
                    self.net.finish(handle.static_out, dict(static_state))
                    torch.cuda.synchronize()
                    rec = _lib.Recorder()
                    with rec:
                        out = self.net.finish(handle.static_out, dict(static_state))
                torch.cuda.synchronize()
                late = handle.late[sig] = _Recorded(rec, static_state, out)
            else:
                _copy_into(late.static_in, state)
                late.replay()
            disps, costs, samples, offs, ranges, info = late.static_out
            out_info = dict(prev_info)
            out_info.update(info)
            return disps, costs, samples, offs, ranges, out_info

    def _run(self, left_feats, right_feats, left_image, right_image, prev_info):
        state = {k: v for k, v in prev_info.items()
                 if k in ("cost_memory", "use_past_cost", "local_map", "local_map_size") and v is not None}
        args = (list(left_feats), list(right_feats), left_image, right_image, state)
        if self.replay == "eager":
            with torch.no_grad():
                return self.net(args[0], args[1], args[2], args[3], dict(prev_info))
        # a replay is tied to what it was recorded on: shapes, (bound) storages, and the caller's stream -- the plan's
        # launches and cross-stream edges name that stream, and the copies into / reads out of the static buffers are
        # ordered with the replay only on it
        sig = _sig_of(args) + ((_ptrs_of(args),) if self.bind else ()) + (torch.cuda.current_stream().cuda_stream,)
        if self.pipeline > 1:                      # double-buffered plans, used alternately
            turn = self._turn.get(sig, 0)
            self._turn[sig] = (turn + 1) % self.pipeline
            sig = sig + (turn,)
            if sig not in self._slot_of:
                self._slot_of[sig] = _new_event_slot()
            self.net.pipeline_slot = self._slot_of[sig]
        cap = self._graphs.get(sig)
        if cap is None:
            cap = self._graphs[sig] = self._record(args) if self.replay == "plan" else self._capture(args)
        if not self.bind:
            _copy_into(cap.static_in, args)
        cap.replay()
        disps, costs, samples, offs, ranges, info = cap.static_out
        out_info = dict(prev_info)
        out_info.update(info)
        return disps, costs, samples, offs, ranges, out_info
