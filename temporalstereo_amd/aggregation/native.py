"""Native inference path of the aggregation pyramid: every stage on the HIP kernels of libts_hip.so.

`NativeAggregator(net)` takes a TEMPORALSTEREO module (eval mode), folds each conv's BatchNorm and
bias into a per-channel (scale, shift), re-lays the weights out for the kernels ([Cin][taps][Cout],
output channel contiguous so one scalar load feeds a whole tap) and then runs the same graph as
aggregation/levels.py with NO framework op on the data path:

  cost volume      ts_block_cost_*                (K1)
  separable convs  ts_conv3d_hw_fwd / ts_conv3d_d_fwd (+ transposed), BN + activation fused   (K3a/b)
  hourglass skips  ts_resize3d_add_act_fwd        (trilinear resize + add + SiLU)            (K3b)
  temporal merge   ts_merge_candidates_fwd        (past_conv + cat + stable sort + gather)   (K3c)
  pyramid fusion   ts_pool3d5_avgmax_fwd + channel-sliced outputs instead of torch.cat       (K3d)
  heads            conv kernels with the tanh-offset epilogue                               (K3e)
  regression       ts_topk_softargmax_fwd         (K4a)
  upsamplers       ts_convex_upsample_fwd / ts_unet_upsample_fwd / ts_deconv2d_k4s2_fwd      (K5)

Semantics are those of the reference modules in eval mode (citations in levels.py / blocks.py).
"""
import os
import time

import torch
import torch.nn as nn

from .. import _lib
from .. import functional as TF

ACT_NONE, ACT_SILU, ACT_RELU, ACT_TANH_OFFSET, ACT_HEAD_PAIR = 0, 1, 2, 3, 4


def _stream():
    return _lib.current_stream_handle()


# ------------------------------------------------------------------------- independent branches
# The pyramid is a chain of small dependent launches; where two sub-chains are independent the shorter
# one can be issued on an auxiliary high-priority stream so that it leaves the critical path.  A
# cross-stream edge costs ~6 us of idle time on this stack, so only the hourglass shortcuts (four
# launches per block) pay for it: measured 1.616 ms/pair with them, 1.708 without any branch, and
# 1.641 / 1.650 / 1.678 when the pooling / the second prediction head / both are branched as well
# (TS_BRANCHES=hourglass,pool,heads re-enables those for experiments).  Off during hipGraph capture;
# NativeAggregator switches it on around a pass.
_PAR = {"on": False, "aux": None, "kinds": frozenset(__import__("os").environ.get("TS_BRANCHES", "hourglass").split(","))}


def _aux_stream():
    """The auxiliary stream of the aggregator whose pass is being issued (set by NativeAggregator.__call__)."""
    return _PAR["aux"]


_STAGE_CAP = int(os.environ.get("TS_STAGE_CHUNK_CAP", "8"))      # K chunk of the f32 kernels while several stages share the chip


def _chunk_cap(cap):
    if cap == 8:
        cap = _STAGE_CAP
    _lib.check(_lib.lib().ts_conv_set_chunk_cap(int(cap)), "ts_conv_set_chunk_cap")


# Cross-stream edges are the currency of the overlap (about 25 per pass), and on MI355X / ROCm 7.2 their cost
# is a property of the stream: of twelve high-priority streams created in a row, nine complete a
# main -> stream -> main round trip (two tiny kernels, two edges) in 31 us and three need 120-150 us -- the same
# streams every time they are measured, normal-priority streams never (tools/exp/edge_latency.py; the runtime
# multiplexes streams onto hardware queues and some placements signal slowly).  A pass whose chain runs on
# such a stream takes 3.0-4.6 ms instead of 1.52 ms.  So streams are qualified once per device by that round
# trip and every aggregator on the device shares the first two that pass.
_QUALIFIED = {}         # device index -> list of qualified high-priority streams
_ROUND_TRIP_SLACK = 1.6


def _round_trip_us(main, s, n=20):
    L = _lib._real_lib()
    buf = torch.zeros(2, 256, device=s.device)
    pm, ps = _lib.ctypes.c_void_p(main.cuda_stream), _lib.ctypes.c_void_p(s.cuda_stream)
    a, b = _lib.ctypes.c_void_p(buf[0].data_ptr()), _lib.ctypes.c_void_p(buf[1].data_ptr())

    def lap(k):
        for _ in range(k):
            for st, other in ((pm, ps), (ps, pm)):
                _lib.check(L.ts_copy_rows_fwd(a, b, 1, 256, 256, 256, st), "ts_copy_rows_fwd")
                _lib.check(L.ts_stream_fork(st, other), "ts_stream_fork")
    lap(5)
    best = None
    for _ in range(3):                          # the minimum of three laps: robust against a busy host
        torch.cuda.synchronize(s.device)
        t0 = time.perf_counter()
        lap(n)
        torch.cuda.synchronize(s.device)
        us = (time.perf_counter() - t0) / n * 1e6
        best = us if best is None else min(best, us)
    return best


def qualified_streams(dev, count, private=False):
    """`count` high-priority streams of `dev` whose cross-stream edges are fast (see above).  Shared ones are the
    same for every caller on the device; private=True draws further qualified streams (several passes in flight)."""
    with torch.cuda.device(dev):
        pool = _QUALIFIED.setdefault(torch.cuda.current_device(), {"shared": [], "floor": None})
        main = torch.cuda.current_stream()
        if pool["floor"] is None:               # the yardstick: a normal-priority stream (never slow in our measurements)
            pool["floor"] = _round_trip_us(main, torch.cuda.Stream(device=dev))

        def draw():
            best = None
            for _ in range(8):
                s = torch.cuda.Stream(device=dev, priority=-1)
                us = _round_trip_us(main, s)
                if us <= _ROUND_TRIP_SLACK * pool["floor"]:
                    return s
                if best is None or us < best[0]:
                    best = (us, s)
            return best[1]                      # none qualified: the least bad one
        if private:
            return [draw() for _ in range(count)]
        while len(pool["shared"]) < count:
            pool["shared"].append(draw())
        return pool["shared"][:count]


def _edge(src, dst):
    """`dst` waits for everything enqueued so far on `src`."""
    rc = _lib.lib().ts_stream_fork(_lib.ctypes.c_void_p(src.cuda_stream), _lib.ctypes.c_void_p(dst.cuda_stream))
    _lib.check(rc, "ts_stream_fork")


class Branch:
    """`with Branch() as br: ...` issues the block on the auxiliary stream, ordered after what the
    current stream has enqueued so far; `br.join()` makes the current stream wait for it.  The
    auxiliary stream is in-order, so one join covers every branch opened before it.
    Allocator note: a branch always starts with an edge from the consumer stream and is joined before
    its operands die, which orders every reuse of a cached block after its last reader."""

    def __init__(self, kind):
        self.on = _PAR["on"] and kind in _PAR["kinds"]
        if self.on:
            self.cur, self.aux = torch.cuda.current_stream(), _aux_stream()

    def __enter__(self):
        if self.on:
            _edge(self.cur, self.aux)
            self.ctx = torch.cuda.stream(self.aux)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.on:
            _edge(self.aux, self.cur)


def _act_code(mod):
    if mod is None:
        return ACT_NONE
    if isinstance(mod, nn.SiLU):
        return ACT_SILU
    if isinstance(mod, nn.ReLU):
        return ACT_RELU
    raise NotImplementedError("activation %r has no fused kernel epilogue" % (mod,))


class Tape:
    """How every kernel-layout array of a NativeAggregator derives from the module's parameters and buffers, so that the arrays
    can be RE-MADE IN PLACE (same storage: recorded launch plans and captured graphs stay valid) by a handful of table-driven
    launches -- ts_conv_weight_layout_many2 for all weight arrays, ts_bn_fold_many for all scale / shift vectors, one split launch
    per bf16-split copy -- instead of ~2,000 framework launches of a rebuild.  Used by InferenceEngine.refresh() after the weights
    changed and, once per step, by train.TrainStep for the previous (eval) frames of a training step.
    Entries are recorded while the aggregator is built (`with tape:` around _build); arrays the tape cannot express raise at
    record time (`unsupported`), and refresh() then reports that a rebuild is needed."""

    def __init__(self):
        self.layouts = []        # (out tensor, out_offset_elems, weight param, A, T, nb, col0, sa, sb, st, out_sa, out_st)
        self.folds = []          # (gamma, beta, mean, var, bias, scale view, shift view, C)
        self.splits = []         # (Folded, "x6" | "x6s", mode)
        self.eps = None
        self.unsupported = []
        self._tables = None

    def __enter__(self):
        global _TAPE
        self._outer, _TAPE = _TAPE, self
        return self

    def __exit__(self, *exc):
        global _TAPE
        _TAPE = self._outer
        return False

    def layout(self, out, weight, transposed, rows=None, col0=0, out_st=None, src_cin0=0, cout_sel=None):
        """out [.., taps, pitch] <- the [Cin][taps][Cout] layout of `weight` ([Cout, Cin, taps...] or, transposed, [Cin, Cout, taps...]).
        rows: (first, count) input channels of the SOURCE written to out's leading rows; col0: first destination column;
        out_st: destination tap pitch (default: out's own); cout_sel: (first, count) output channels of the source."""
        w = weight
        d0, d1 = w.shape[0], w.shape[1]
        T = w[0, 0].numel()
        cout, cin = (d1, d0) if transposed else (d0, d1)
        s_ci, s_co = ((d1 * T, T) if transposed else (T, d1 * T))
        c0, nci = rows if rows is not None else (0, cin)
        o0, nco = cout_sel if cout_sel is not None else (0, cout)
        self.layouts.append(dict(out=out, w=w, A=nci, T=T, nb=nco, col0=col0, sa=s_ci, sb=s_co, st=1, src_off=(c0 + src_cin0) * s_ci + o0 * s_co,
                                 out_sa=out.stride(0), out_st=out.stride(1) if out_st is None else out_st))
        self._tables = None

    def fold(self, scale, shift, bias, bn, cout):
        if bn is not None:
            if self.eps is not None and float(bn.eps) != self.eps:
                self.unsupported.append("BatchNorm layers with different eps")
            self.eps = float(bn.eps)
        self.folds.append(dict(scale=scale, shift=shift, bias=bias, bn=bn, C=cout))
        self._tables = None

    def _upload(self, dev):
        import numpy as np
        lay = np.zeros(len(self.layouts), dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("A", "<i4"), ("T", "<i4"), ("nb", "<i4"), ("col0", "<i4"),
                                                          ("sa", "<i8"), ("sb", "<i8"), ("st", "<i8"), ("osa", "<i8"), ("ost", "<i8"),
                                                          ("flip", "<i4"), ("reserved", "<i4")]))
        most = 1
        for i, e in enumerate(self.layouts):
            w = e["w"]
            if not w.is_contiguous():
                raise RuntimeError("Tape: a source weight is not contiguous")
            lay[i] = (w.data_ptr() + 4 * e["src_off"], e["out"].data_ptr(), e["A"], e["T"], e["nb"], e["col0"], e["sa"], e["sb"], e["st"],
                      e["out_sa"], e["out_st"], 0, 0)
            most = max(most, e["A"] * e["T"] * e["nb"])
        fo = np.zeros(len(self.folds), dtype=np.dtype([("gamma", "<u8"), ("beta", "<u8"), ("mean", "<u8"), ("var", "<u8"), ("bias", "<u8"),
                                                       ("scale", "<u8"), ("shift", "<u8"), ("C", "<i4"), ("pad", "<i4")]))
        P = lambda t: t.data_ptr() if t is not None else 0
        for i, e in enumerate(self.folds):
            bn = e["bn"]
            if bn is not None:
                g, b = (bn.weight, bn.bias) if bn.affine else (None, None)
                fo[i] = (P(g), P(b), P(bn.running_mean), P(bn.running_var), P(e["bias"]), P(e["scale"]), P(e["shift"]), e["C"], e["C"])
            else:
                fo[i] = (0, 0, 0, 0, P(e["bias"]), P(e["scale"]), P(e["shift"]), e["C"], e["C"])
        up = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(dev)
        self._tables = (up(lay), up(fo), min(64, (most + 255) // 256))

    def refresh(self, dev):
        """Re-make every recorded array from the CURRENT parameters / running statistics, in place, on the current stream."""
        if self.unsupported:
            raise RuntimeError("Tape.refresh: %s -- rebuild the aggregator instead" % "; ".join(self.unsupported))
        L = _lib.lib()
        if self._tables is None:
            self._upload(dev)
        lay, fo, blocks = self._tables
        if self.layouts:
            _lib.check(L.ts_conv_weight_layout_many2(_lib.ptr(lay), len(self.layouts), blocks, _stream()), "ts_conv_weight_layout_many2")
        if self.folds:
            _lib.check(L.ts_bn_fold_many(_lib.ptr(fo), len(self.folds), self.eps if self.eps is not None else 1e-5, _stream()), "ts_bn_fold_many")
        for f, kind, mode in self.splits:
            if kind == "x6":
                _lib.check(L.ts_conv3d_hw_x6_weight_split(f.w.data_ptr(), f.w6.data_ptr(), f.cin, f.cout, _stream()), "ts_conv3d_hw_x6_weight_split")
            else:
                _lib.check(L.ts_conv3d_hw_x6s_weight_split(f.w.data_ptr(), f.w6s.data_ptr(), f.cin, f.cout, f.w.shape[2], mode, _stream()),
                           "ts_conv3d_hw_x6s_weight_split")


_TAPE = None


class Folded:
    """Kernel-ready form of one conv (+BatchNorm (+activation))."""

    def __init__(self, weight, bias, bn, act, transposed, kind):
        self._tape = _TAPE
        self.src = (weight, bias, bn, transposed)
        w = weight.detach().float()
        if transposed:                      # ConvTranspose: [Cin, Cout, ...] -> [Cout, Cin, ...]
            w = w.transpose(0, 1)
        cout, cin = w.shape[0], w.shape[1]
        taps = w.shape[2:].numel()
        self.cin, self.cout, self.kind, self.act = cin, cout, kind, act
        self.kshape = tuple(w.shape[2:])
        pad = int(_lib.lib().ts_conv_cout_pad(cout))      # every entry, ts_deconv2d_k4s2_fwd included, takes this pitch (8 | 16 | 32 | 64 ...)
        if pad < cout:
            raise NotImplementedError("Cout=%d has no kernel bucket" % cout)
        wt = torch.zeros(cin, taps, pad, device=w.device, dtype=torch.float32)
        wt[:, :, :cout] = w.reshape(cout, cin, taps).permute(1, 2, 0)
        self.w = wt.contiguous()
        scale = torch.ones(pad, device=w.device)
        shift = torch.zeros(pad, device=w.device)
        b = bias.detach().float() if bias is not None else torch.zeros(cout, device=w.device)
        if bn is not None:
            # any running-statistics BatchNorm folds exactly in eval mode, nn.SyncBatchNorm (what dist.sync_batchnorm /
            # Lightning's sync_batchnorm=True leave behind after data-parallel training) included
            if not isinstance(bn, nn.modules.batchnorm._BatchNorm) or bn.running_mean is None:
                raise NotImplementedError("only BatchNorm with running statistics can be folded, got %r" % (bn,))
            gamma = bn.weight.detach().float() if bn.affine else torch.ones(cout, device=w.device)
            beta = bn.bias.detach().float() if bn.affine else torch.zeros(cout, device=w.device)
            s = gamma / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            scale[:cout] = s
            shift[:cout] = beta + (b - bn.running_mean.detach().float()) * s
        else:
            shift[:cout] = b
        self.scale, self.shift = scale.contiguous(), shift.contiguous()
        if _TAPE is not None:
            if weight.dtype != torch.float32:
                _TAPE.unsupported.append("non-fp32 weights")
            _TAPE.layout(self.w, weight.detach(), transposed)
            _TAPE.fold(self.scale, self.shift, bias, bn, cout)


def fold_split_input(f, n):
    """(head, rest): `head` is the raw convolution over input channels [:n] (no bias / BatchNorm /
    activation), `rest` the layer over channels [n:] with the original epilogue, so that
    layer(x) == rest(x[:, n:], addend=head(x[:, :n]))."""
    head, rest = object.__new__(Folded), object.__new__(Folded)
    for g, lo, hi in ((head, 0, n), (rest, n, f.cin)):
        g.cin, g.cout, g.kind, g.kshape = hi - lo, f.cout, f.kind, f.kshape
        g._tape = getattr(f, "_tape", None)
        g.w = f.w[lo:hi].contiguous()          # a leading-rows slice is contiguous: a VIEW of f.w (an in-place re-fold of f reaches it)
        assert g.w.data_ptr() == f.w[lo:hi].data_ptr()
    head.act, rest.act = ACT_NONE, f.act
    head.scale, head.shift = torch.ones_like(f.scale), torch.zeros_like(f.shift)
    rest.scale, rest.shift = f.scale, f.shift
    return head, rest


def fold_concat(a, b):
    """One Folded computing [a | b] along Cout (same input, same taps, same activation)."""
    if (a.cin, a.kind, a.act, a.kshape) != (b.cin, b.kind, b.act, b.kshape):
        raise ValueError("fold_concat: layers differ in more than their output channels")
    f = object.__new__(Folded)
    f.cin, f.cout, f.kind, f.act, f.kshape = a.cin, a.cout + b.cout, a.kind, a.act, a.kshape
    pad = int(_lib.lib().ts_conv_cout_pad(f.cout))
    if pad < f.cout:
        raise NotImplementedError("Cout=%d has no kernel bucket" % f.cout)
    taps = a.w.shape[1]
    f.w = torch.zeros(a.cin, taps, pad, device=a.w.device, dtype=torch.float32)
    f.w[:, :, :a.cout] = a.w[:, :, :a.cout]
    f.w[:, :, a.cout:f.cout] = b.w[:, :, :b.cout]
    f.scale = torch.ones(pad, device=a.w.device)
    f.shift = torch.zeros(pad, device=a.w.device)
    f.scale[:a.cout] = a.scale[:a.cout]; f.scale[a.cout:f.cout] = b.scale[:b.cout]
    f.shift[:a.cout] = a.shift[:a.cout]; f.shift[a.cout:f.cout] = b.shift[:b.cout]
    f._tape = getattr(a, "_tape", None)
    if f._tape is not None:
        col = 0
        for g in (a, b):
            src = getattr(g, "src", None)
            if src is None:
                f._tape.unsupported.append("concatenation of derived layers")
                break
            weight, bias, bn, transposed = src
            f._tape.layout(f.w, weight.detach(), transposed, col0=col)
            f._tape.fold(f.scale[col:col + g.cout], f.shift[col:col + g.cout], bias, bn, g.cout)
            col += g.cout
    return f


def fold_wrapper(m, kind, transposed=False):
    """Conv3d / Conv2d / ConvTranspose wrappers of layers.py (conv -> .norm -> .activation)."""
    return Folded(m.weight, m.bias, getattr(m, "norm", None), _act_code(getattr(m, "activation", None)), transposed, kind)


# ------------------------------------------------------------------------------------------- ops
def _strides5(t):
    if t.stride(4) != 1 or t.stride(3) != t.shape[4] or t.stride(2) != t.shape[3] * t.shape[4]:
        raise ValueError("tensor planes must be dense (channel-sliced views of contiguous buffers are fine)")
    return t.stride(0), t.stride(1)


# TS_CONV_X6=0: every (1,3,3) convolution on the f32-input MFMA kernel (A/B measurements, bit-exact fp32 products)
X6 = os.environ.get("TS_CONV_X6", "1") != "0"
# Grids below this many x6 workgroups stay on the f32 kernel (with its split-K) -- unless the x6 kernel splits the reduction itself
# (round 3: long reductions on small grids, ts_conv3d_hw_x6_workspace_bytes > 0).  Measured 32 ... 384: 128 best at batch 1, flat at 4;
# layer by layer: tools/exp/x6_splitk_bench.py.
_X6_MIN_GRID = int(os.environ.get("TS_CONV_X6_MIN_GRID", "128"))
# ... and where the f32 kernel would not split its reduction (Cin < 64), from this many x6 workgroups
_X6_MIN_GRID_UNSPLIT = int(os.environ.get("TS_CONV_X6_MIN_GRID_UNSPLIT", "64"))


def x6_weights(f):
    """The bf16-split copy of a Folded layer's weights (ts_conv3d_hw_x6_weight_split), made on first use -- outside any plan
    recording: the split runs once, replays only read it."""
    w6 = getattr(f, "w6", None)
    if w6 is None:
        L = _lib._real_lib()
        w6 = torch.empty(int(L.ts_conv3d_hw_x6_weight_bytes(f.cin, f.cout)), device=f.w.device, dtype=torch.uint8)
        _lib.check(L.ts_conv3d_hw_x6_weight_split(f.w.data_ptr(), w6.data_ptr(), f.cin, f.cout, _stream()), "ts_conv3d_hw_x6_weight_split")
        f.w6 = w6
        if getattr(f, "_tape", None) is not None:
            f._tape.splits.append((f, "x6", 0))
    return w6


# TS_CONV_X6S=0: stride-2 / transposed convolutions and the UNet deconvolutions stay on the f32-input MFMA kernels (A/B measurements)
X6S = os.environ.get("TS_CONV_X6S", "1") != "0"      # (in effect only with X6)
# grids below this many x6s workgroups stay on the f32 kernel's 64-pixel tiles (the small layers of the hourglasses: launch-bound)
# (tools/x6s_bench.py, x6s / f32: 32 -> 64 stride 2 on 2 x 272 x 480 38 / 53 us, 16 -> 32 on 5 x 136 x 240 11.6 / 13.8, 32 -> 64 on 12 x 32 x 64 -- 96 workgroups
# -- 12.7 / 11.9; deconvolutions 32 -> 32 on 136 x 240 21.4 / 23.1, 32 -> 9 on 272 x 480 23.3 / 33.5; batch 4: 1.16 - 1.41 x)
_X6S_MIN_GRID = int(os.environ.get("TS_CONV_X6S_MIN_GRID", "256"))
# the (1,3,3)^T layers are the hourglasses' small ones: 16 -> 8 on 3 x 68 x 120 (108 workgroups) 15.5 vs 11.7 us, batch 4 (432) 19.9 vs 21.9
_X6S_MIN_GRID_T3 = int(os.environ.get("TS_CONV_X6S_MIN_GRID_T3", "400"))
X6S_S2, X6S_T3, X6S_T4 = 0, 1, 2


def x6s_weights(f, mode):
    """bf16-split weights of a strided / transposed layer in the 8-channel-chunk layout of csrc/conv_x6s.hip, made on first use
    (outside any plan recording, like x6_weights)."""
    w6 = getattr(f, "w6s", None)
    if w6 is None:
        L = _lib._real_lib()
        w6 = torch.empty(int(L.ts_conv3d_hw_x6s_weight_bytes(f.cin, f.cout, mode)), device=f.w.device, dtype=torch.uint8)
        _lib.check(L.ts_conv3d_hw_x6s_weight_split(f.w.data_ptr(), w6.data_ptr(), f.cin, f.cout, f.w.shape[2], mode, _stream()),
                   "ts_conv3d_hw_x6s_weight_split")
        f.w6s = w6
        f.w6s_mode = mode
        if getattr(f, "_tape", None) is not None:
            f._tape.splits.append((f, "x6s", mode))
    elif getattr(f, "w6s_mode", mode) != mode:
        raise RuntimeError("x6s_weights: this layer's weights were split for mode %d, asked for mode %d" % (f.w6s_mode, mode))
    return w6


def x6s_grid(B, cout, D, H, W, mode):
    if mode == X6S_S2:
        return ((((H - 1) // 2 + 1) + 3) // 4) * ((((W - 1) // 2 + 1) + 31) // 32) * D * B * ((cout + 31) // 32)
    return ((H + 7) // 8) * ((W + 31) // 32) * D * B * ((cout + 15) // 16)


def conv_hw(x, f, stride=1, dilation=1, transposed=False, out=None, act=None, act_param=0.0, addend=None, second=None,
            out_second=None):
    """x [B,Cin,D,H,W] -> [B,Cout,D,Ho,Wo].  addend [B,Cout,1,Ho,Wo]: added to every depth plane's raw sum.
    second: another tensor of x's shape with B == 1 -- the two are convolved as ONE batch of two (the batch stride
    handed to the kernel is simply the distance between the two allocations), without stacking them first.
    out_second: with `second`, the second batch element's output goes to this separate tensor (`out` holds the first)."""
    B, Cin, D, H, W = x.shape
    if second is not None:
        if B != 1 or second.shape != x.shape or _strides5(second)[1] != _strides5(x)[1]:
            raise ValueError("conv_hw(second=...): two tensors of one shape and channel stride with batch 1")
        if (second.data_ptr() - x.data_ptr()) % 4:
            raise ValueError("conv_hw(second=...): allocations are not 4-byte aligned relative to each other")
        B = 2
    assert Cin == f.cin, (Cin, f.cin)
    if transposed:
        Ho, Wo = 2 * H, 2 * W
    else:
        Ho = (H + 2 * dilation - 2 * dilation - 1) // stride + 1
        Wo = (W + 2 * dilation - 2 * dilation - 1) // stride + 1
    if out is None:
        out = torch.empty((B, f.cout, D, Ho, Wo), device=x.device, dtype=torch.float32)
    ib, ic = _strides5(x)
    if second is not None:
        ib = (second.data_ptr() - x.data_ptr()) // 4
        _lib.ptr(second)                                   # kept alive with a recorded plan
    ob, oc = _strides5(out)
    if out_second is not None:
        if second is None or out.shape[0] != 1 or out_second.shape != out.shape or _strides5(out_second)[1] != oc:
            raise ValueError("conv_hw(out_second=...): needs `second` and two outputs of one shape and channel stride")
        ob = (out_second.data_ptr() - out.data_ptr()) // 4
        _lib.ptr(out_second)
    L = _lib.lib()
    if X6 and X6S and addend is None and (stride == 2 or transposed) and dilation == 1:
        mode = X6S_T3 if transposed else X6S_S2
        if L.ts_conv3d_hw_x6s_supported(Cin, f.cout, H, W, mode) and x6s_grid(B, f.cout, D, H, W, mode) >= (_X6S_MIN_GRID_T3 if transposed else _X6S_MIN_GRID):
            rc = L.ts_conv3d_hw_x6s_fwd(_lib.ptr(x), _lib.ptr(x6s_weights(f, mode)), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                                        B, Cin, f.cout, D, H, W, mode, f.act if act is None else act, float(act_param),
                                        ib, ic, ob, oc, _stream())
            _lib.check(rc, "ts_conv3d_hw_x6s_fwd")
            return out
    wsb = int(L.ts_conv3d_hw_workspace_bytes(B, Cin, f.cout, D, H, W, stride, int(transposed)))
    x6_grid = ((H + 7) // 8) * ((W + 31) // 32) * D * B * ((f.cout + 31) // 32)
    x6_ok = X6 and L.ts_conv3d_hw_x6_supported(Cin, f.cout, W, stride, dilation, int(transposed))
    wsb6 = int(L.ts_conv3d_hw_x6_workspace_bytes(B, Cin, f.cout, D, H, W)) if x6_ok else 0
    if x6_ok and Cin < 32 and x6_grid < 512:
        x6_ok = False        # one K chunk on a small grid: the f32 kernel on 64-pixel tiles is the shorter launch (16 -> 16 on 5 x 68 x 120: 7.1 vs 8.7 us)
    if x6_ok and (x6_grid >= _X6_MIN_GRID or wsb6 or (not wsb and x6_grid >= _X6_MIN_GRID_UNSPLIT)):
        # fp32 products from bf16 pieces on the bf16 matrix pipe (max error below the f32 kernel's: DESIGN.md section 4; 3/8 of the matrix time);
        # small grids stay on the f32 kernel (64-pixel tiles, its own split-K) unless the reduction is long enough for the x6 kernel's
        # split-K (wsb6 > 0) -- also where the f32 kernel would not split: 32 -> 32 on 3 x 34 x 60 is 6.2 us there against 15.7 us on x6
        ws6 = torch.empty(wsb6, device=x.device, dtype=torch.uint8) if wsb6 else None
        rc = L.ts_conv3d_hw_x6_fwd(_lib.ptr(x), _lib.ptr(x6_weights(f)), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                                   B, Cin, f.cout, D, H, W, dilation, f.act if act is None else act, float(act_param),
                                   ib, ic, ob, oc, _lib.ptr(addend), addend.stride(0) if addend is not None else 0, _lib.ptr(ws6), wsb6,
                                   _stream())
        _lib.check(rc, "ts_conv3d_hw_x6_fwd")
        return out
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8) if wsb else None
    rc = L.ts_conv3d_hw_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                            B, Cin, f.cout, D, H, W, stride, dilation, int(transposed),
                            f.act if act is None else act, float(act_param), ib, ic, ob, oc,
                            _lib.ptr(addend), addend.stride(0) if addend is not None else 0, _lib.ptr(ws), wsb, _stream())
    _lib.check(rc, "ts_conv3d_hw_fwd")
    return out


def conv_d(x, f, k, stride=1, dilation=1, padding=0, transposed=False, out=None, act=None, act_param=0.0):
    """x [B,Cin,Din,H,W] -> [B,Cout,Dout,H,W]."""
    B, Cin, Din, H, W = x.shape
    assert Cin == f.cin, (Cin, f.cin)
    Dout = 2 * Din if transposed else (Din + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((B, f.cout, Dout, H, W), device=x.device, dtype=torch.float32)
    ib, ic = _strides5(x)
    ob, oc = _strides5(out)
    rc = _lib.lib().ts_conv3d_d_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                                    B, Cin, f.cout, Din, H, W, k, stride, dilation, padding, int(transposed),
                                    f.act if act is None else act, float(act_param), ib, ic, ob, oc, _stream())
    _lib.check(rc, "ts_conv3d_d_fwd")
    return out


# TS_FUSED_K1=0: the sampled levels build the warped half of their volume and convolve it (rounds 1-3) instead of the
# pre-contracted form below (A/B measurements; both are held to the same fixtures)
FUSED_K1 = os.environ.get("TS_FUSED_K1", "1") != "0"


def block_cost_corr(left, right, disp, scales):
    """The correlation blocks of the sampled block_cost alone (functional.block_cost_corr; reached as TF.<name> so that bench.py's
    K1 probe sees the call)."""
    return TF.block_cost_corr(left, right, disp, scales)


def conv_hw_warp(corr, f, q, disp, base, dilation=1):
    """First layer of a sampled level from the correlation blocks, the pre-contracted right map `q` and the candidates
    (ts_conv3d_hw_warp_fwd): [B, Cout, D, H, W]."""
    B, Cc, D, H, W = corr.shape
    assert Cc == f.cin, (Cc, f.cin)
    out = torch.empty((B, f.cout, D, H, W), device=corr.device, dtype=torch.float32)
    ib, ic = _strides5(corr)
    ob, oc = _strides5(out)
    wsb = TF._q("ts_conv3d_hw_warp_workspace_bytes", B, f.cout, D, H, W)
    ws = torch.empty(wsb, device=corr.device, dtype=torch.uint8)
    rc = _lib.lib().ts_conv3d_hw_warp_fwd(_lib.ptr(corr), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                                          _lib.ptr(q), _lib.ptr(disp), _lib.ptr(base), B, Cc, f.cout, D, H, W, dilation, f.act, 0.0,
                                          ib, ic, ob, oc, base.stride(0) if base is not None else 0, _lib.ptr(ws), wsb, _stream())
    _lib.check(rc, "ts_conv3d_hw_warp_fwd")
    return out


def resize_add_act(a, add, size, act=ACT_SILU):
    B, C, Da, Ha, Wa = a.shape
    D, H, W = size
    out = torch.empty((B, C, D, H, W), device=a.device, dtype=torch.float32)
    ab, ac = _strides5(a)
    bb, bc = _strides5(add) if add is not None else (0, 0)
    ob, oc = _strides5(out)
    rc = _lib.lib().ts_resize3d_add_act_fwd(_lib.ptr(a), _lib.ptr(add), _lib.ptr(out), B, C, Da, Ha, Wa, D, H, W, act,
                                            ab, ac, bb, bc, ob, oc, _stream())
    _lib.check(rc, "ts_resize3d_add_act_fwd")
    return out


def pool5(x, out_avg, out_max):
    B, C, D, H, W = x.shape
    xb, xc = _strides5(x); ab, ac = _strides5(out_avg); mb, mc = _strides5(out_max)
    rc = _lib.lib().ts_pool3d5_avgmax_fwd(_lib.ptr(x), _lib.ptr(out_avg), _lib.ptr(out_max), B, C, D, H, W,
                                          xb, xc, ab, ac, mb, mc, _stream())
    _lib.check(rc, "ts_pool3d5_avgmax_fwd")


def resize_bilinear(x, size, value_scale=1.0, out=None):
    """F.interpolate(x * value_scale, size, 'bilinear', align_corners=True); `out` may be a channel slice."""
    B, C, h, w = x.shape
    if out is None:
        out = torch.empty((B, C, size[0], size[1]), device=x.device, dtype=torch.float32)
    x = _lib.contiguous(x)                     # a named reference: a temporary's block could be re-used by a later argument's allocation
    rc = _lib.lib().ts_resize_bilinear_fwd(_lib.ptr(x), _lib.ptr(out), B, C, h, w, size[0], size[1],
                                           float(value_scale), out.stride(0), _stream())
    _lib.check(rc, "ts_resize_bilinear_fwd")
    return out


def resize_bilinear_pair(x0, x1, size, scale0, scale1):
    """Two same-shaped maps through resize_bilinear in one launch."""
    B, C, h, w = x0.shape
    if x1.shape != x0.shape:
        raise RuntimeError("resize_bilinear_pair: shapes differ")
    o0 = torch.empty((B, C, size[0], size[1]), device=x0.device, dtype=torch.float32)
    o1 = torch.empty_like(o0)
    x0, x1 = _lib.contiguous(x0), _lib.contiguous(x1)
    rc = _lib.lib().ts_resize_bilinear_pair_fwd(_lib.ptr(x0), _lib.ptr(x1), _lib.ptr(o0), _lib.ptr(o1),
                                                B, C, h, w, size[0], size[1], float(scale0), float(scale1), _stream())
    _lib.check(rc, "ts_resize_bilinear_pair_fwd")
    return o0, o1


def copy_rows(src, dst):
    """dst[...] = src for a `dst` that is a channel slice of a larger contiguous tensor ([B, C, ...])."""
    B = src.shape[0]
    n = src[0].numel()
    src = _lib.contiguous(src)
    rc = _lib.lib().ts_copy_rows_fwd(_lib.ptr(src), _lib.ptr(dst), B, n, n, dst.stride(0), _stream())
    _lib.check(rc, "ts_copy_rows_fwd")


def range_candidates(disp, rng, extra_front=0):
    """(low, high, candidates [B, extra_front + 5, H, W]) of the next level from an upsampled disparity;
    the first `extra_front` candidate planes are left for the caller (local-map candidates)."""
    B, _, H, W = disp.shape
    low = torch.empty_like(disp)
    high = torch.empty_like(disp)
    cand = torch.empty((B, extra_front + 5, H, W), device=disp.device, dtype=torch.float32)
    rc = _lib.lib().ts_range_candidates_fwd(_lib.ptr(disp), _lib.ptr(low), _lib.ptr(high), _lib.ptr(cand), B, H, W,
                                            float(rng), extra_front, extra_front + 5, _stream())
    _lib.check(rc, "ts_range_candidates_fwd")
    return low, high, cand


def split_sampled_first_layer(f0, scales):
    """The (1,3,3) layer over a sampled level's volume [left x D | warped right | corr] (precise.py:88-91, fine.py:96-103) in
    the pieces the native pipeline runs -> (left, rest, corr, q):
      left  the raw convolution over the first C channels: they are the left features repeated over D (block_cost.py:51), so this
            part is computed once per pixel and handed on as a D-invariant addend;
      rest  the layer over [warped | corr] with the original epilogue (the form of rounds 1-3: ts_block_cost_sampled_warped_fwd);
      corr  the layer over the correlation blocks alone, and
      q     the 1x1 convolution right -> Q [9*Cout planes, plane = tap * Cout + co]: the warp is a two-tap interpolation whose
            weights do not depend on the channel, so it commutes with the layer's channel contraction (ts_conv3d_hw_warp_fwd)."""
    C = f0.cin * 8 // (16 + scales)                  # cin == 2C + scales * C/8
    if 2 * C + scales * (C // 8) != f0.cin or f0.kshape != (1, 3, 3):
        raise NotImplementedError("first aggregation layer is not a (1,3,3) conv over [left | warped | corr]")
    left, rest = fold_split_input(f0, C)
    _, corr = fold_split_input(rest, C)
    q = object.__new__(Folded)
    q.cin, q.cout, q.kind, q.act, q.kshape = C, 9 * rest.cout, "d", ACT_NONE, (1, 1, 1)
    pad = int(_lib.lib().ts_conv_cout_pad(q.cout))
    q.w = torch.zeros(C, 1, pad, device=rest.w.device, dtype=torch.float32)
    q.w[:, 0, :q.cout] = rest.w[:C, :, :rest.cout].reshape(C, 9 * rest.cout)         # [c][tap][co] -> plane tap * Cout + co
    q.scale, q.shift = torch.ones(pad, device=q.w.device), torch.zeros(pad, device=q.w.device)
    q._tape = getattr(f0, "_tape", None)
    if q._tape is not None:
        src = getattr(f0, "src", None)
        if src is None or src[3]:
            q._tape.unsupported.append("first layer of a sampled level is itself derived")
        else:       # q.w[c][0][tap * Cout + co] = W[co][C + c][tap]: rows = the warped half's input channels, tap pitch = Cout
            q._tape.layout(q.w, src[0].detach(), False, rows=(C, C), out_st=rest.cout)
    return left, rest, corr, q


# ------------------------------------------------------------------------------------------- blocks
class SepConv:
    """DepthwiseConv3D / DepthwiseConvTranspose3D (blocks.py) in kernel-ready form."""

    def __init__(self, mod, transposed=False):
        c0, c1 = mod.conv[0], mod.conv[1]
        self.transposed = transposed
        self.f0 = fold_wrapper(c0, "hw", transposed)
        self.f1 = fold_wrapper(c1, "d", transposed)
        self.k = c1.kernel_size[0]
        self.stride, self.dil = c0.stride[1], c0.dilation[1]
        self.pad_d = c1.padding[0]
        if c0.kernel_size != (1, 3, 3) or c0.padding[1] != self.dil or c1.stride[0] != self.stride:
            raise NotImplementedError("separable conv outside the (1,3,3)+(k,1,1) family: %r" % (mod,))

    def __call__(self, x, out=None, addend=None):
        y = conv_hw(x, self.f0, self.stride, self.dil, self.transposed, addend=addend)
        return conv_d(y, self.f1, self.k, self.stride, self.dil, self.pad_d, self.transposed, out=out)


class Hourglass:
    """ResidualBlock3D.forward (module.py:272-297)."""

    def __init__(self, mod):
        self.c1, self.c2, self.c3, self.c4 = (SepConv(getattr(mod, n)) for n in ("conv1", "conv2", "conv3", "conv4"))
        self.c5, self.c6 = SepConv(mod.conv5, True), SepConv(mod.conv6, True)
        self.s5, self.s6 = SepConv(mod.shortcut5), SepConv(mod.shortcut6)
        # conv4 has no activation of its own; the SiLU that follows it (:281) is fused into its last kernel
        self.c4.f1.act = ACT_SILU

    def __call__(self, x):
        pre = self.c2(self.c1(x))
        br = Branch("hourglass")
        with br:                                             # both shortcuts: four launches off the critical path,
            s6 = self.s6(x)                                  # behind ONE cross-stream edge (each edge costs ~6 us)
            s5 = self.s5(pre)
        out = self.c5(self.c4(self.c3(pre)))
        br.join()
        out = resize_add_act(out, s5, pre.shape[-3:])
        return resize_add_act(self.c6(out), s6, x.shape[-3:])


class Heads:
    """PredictionHeads (module.py:356-398): (cost, off) [B,D,H,W]."""

    def __init__(self, mod):
        self.delta = float(mod.delta)
        c0, self.c1 = fold_wrapper(mod.cost_head[0], "d"), fold_wrapper(mod.cost_head[1], "hw")
        o0, self.o1 = fold_wrapper(mod.off_head[0], "d"), fold_wrapper(mod.off_head[1], "hw")
        self.C = C = c0.cout
        self.co0 = fold_concat(c0, o0)          # both heads start with a (3,1,1) conv of the same input: one launch
        # ... and end with a C -> 1 (1,3,3) conv each: one block-diagonal 2C -> 2 convolution (channel 0 reads the
        # first C inputs, channel 1 the last C), the offset head's tanh applied to channel 1 only (ACT_HEAD_PAIR)
        f = object.__new__(Folded)
        f.cin, f.cout, f.kind, f.act, f.kshape = 2 * C, 2, "hw", ACT_HEAD_PAIR, self.c1.kshape
        pad = int(_lib.lib().ts_conv_cout_pad(2))
        f.w = torch.zeros(2 * C, self.c1.w.shape[1], pad, device=self.c1.w.device, dtype=torch.float32)
        f.w[:C, :, 0] = self.c1.w[:, :, 0]
        f.w[C:, :, 1] = self.o1.w[:, :, 0]
        f.scale = torch.ones(pad, device=f.w.device); f.shift = torch.zeros(pad, device=f.w.device)
        f.scale[0], f.scale[1] = self.c1.scale[0], self.o1.scale[0]
        f.shift[0], f.shift[1] = self.c1.shift[0], self.o1.shift[0]
        f._tape = _TAPE
        if _TAPE is not None:
            for k, (m, g) in enumerate(((mod.cost_head[1], self.c1), (mod.off_head[1], self.o1))):
                _TAPE.layout(f.w[k * C:(k + 1) * C], m.weight.detach(), False, col0=k)          # rows [kC, (k+1)C), column k
                _TAPE.fold(f.scale[k:k + 1], f.shift[k:k + 1], m.bias, getattr(m, "norm", None), 1)
        self.pair = f

    def __call__(self, x):
        y = conv_d(x, self.co0, 3, 1, 1, 1)
        B, _, D, H, W = y.shape
        both = torch.empty((2, B, D, H, W), device=y.device, dtype=torch.float32)      # [cost | off], each [B,D,H,W] dense
        conv_hw(y, self.pair, out=both.permute(1, 0, 2, 3, 4), act_param=self.delta)
        return both[0], both[1]


class ConvexUp:
    """ConvexUpsample (module.py:300-353) with mask.0+mask.1 (conv+BN) folded."""

    def __init__(self, mod):
        self.r, self.k = mod.upscale_factor, mod.window_size
        if self.k != 3:
            raise NotImplementedError("window_size != 3")
        self.m0 = Folded(mod.mask[0].weight.unsqueeze(2), mod.mask[0].bias, mod.mask[1], ACT_SILU, False, "hw")
        w3 = mod.mask[3].weight
        self.m3 = Folded(w3.reshape(w3.shape[0], w3.shape[1], 1, 1, 1), mod.mask[3].bias, None, ACT_NONE, False, "d")

    def mask(self, feat):
        """The 9 * r^2 convex-combination logits: a function of the left features only."""
        return conv_d(conv_hw(feat.unsqueeze(2), self.m0), self.m3, 1)

    def with_candidates(self, feat, disp, m, rng, extra_front=0):
        """(upsampled disparity, low, high, candidates) of the next level in one launch."""
        B, _, H, W = disp.shape
        if m is None:
            m = self.mask(feat)
        Ho, Wo = H * self.r, W * self.r
        out = torch.empty((B, 1, Ho, Wo), device=disp.device, dtype=torch.float32)
        low, high = torch.empty_like(out), torch.empty_like(out)
        cand = torch.empty((B, extra_front + 5, Ho, Wo), device=disp.device, dtype=torch.float32)
        disp = _lib.contiguous(disp)
        rc = _lib.lib().ts_convex_upsample_candidates_fwd(_lib.ptr(m), _lib.ptr(disp), _lib.ptr(out), _lib.ptr(low),
                                                          _lib.ptr(high), _lib.ptr(cand), B, H, W, self.r, float(self.r), float(rng),
                                                          extra_front, extra_front + 5, _stream())
        _lib.check(rc, "ts_convex_upsample_candidates_fwd")
        return out, low, high, cand

    def __call__(self, feat, disp, m=None):
        B, _, H, W = disp.shape
        if m is None:
            m = self.mask(feat)
        out = torch.empty((B, 1, H * self.r, W * self.r), device=disp.device, dtype=torch.float32)
        disp = _lib.contiguous(disp)
        rc = _lib.lib().ts_convex_upsample_fwd(_lib.ptr(m), _lib.ptr(disp), _lib.ptr(out), B, H, W, self.r,
                                               float(self.r), _stream())
        _lib.check(rc, "ts_convex_upsample_fwd")
        return out


class _LevelBase:
    def __init__(self, mod):
        self.mod = mod
        self.C, self.topk, self.scales = mod.C, mod.topk, mod.block_cost_scale
        self.init0, self.hg, self.init2 = SepConv(mod.init3d[0]), Hourglass(mod.init3d[1]), SepConv(mod.init3d[2])
        self.heads = Heads(mod.pred_heads)

    def init3d(self, raw, addend=None):
        return self.init2(self.hg(self.init0(raw, addend=addend)))

    def split_reference_half(self):
        """Sampled levels: see split_sampled_first_layer."""
        if self.init0.stride != 1 or self.init0.transposed:
            raise NotImplementedError("first aggregation layer is not a stride-1 conv over [left | warped | corr]")
        self.init0_left, self.init0.f0, self.init0_corr, self.init0_q = split_sampled_first_layer(self.init0.f0, self.scales)

    def left_term(self, left):
        return conv_hw(left.unsqueeze(2), self.init0_left, 1, self.init0.dil)

    def right_term(self, right):
        """Q [B, 9*Cout, H, W]: the warped half of the first layer contracted over its channels BEFORE the warp (a function of
        the right map only: issued with the other feature-only work at the start of a pass)."""
        return conv_d(right.unsqueeze(2), self.init0_q, 1).squeeze(2)

    def first_layer_fused(self, left, right, ds, lterm, rterm):
        """init3d[0]'s (1,3,3) half on [corr | Q] instead of on the [warped | corr] volume, then its (k,1,1) half."""
        corr = block_cost_corr(left, right, ds, self.scales)
        i0 = self.init0
        y = conv_hw_warp(corr, self.init0_corr, rterm, ds, lterm.squeeze(2) if lterm.dim() == 5 else lterm, i0.dil)
        return conv_d(y, i0.f1, i0.k, i0.stride, i0.dil, i0.pad_d, i0.transposed)

    def init3d_from(self, first):
        return self.init2(self.hg(first))

    def feature_terms(self, left, right):
        """(left term, right term) of the first layer: functions of the feature maps only."""
        return self.left_term(left), (self.right_term(right) if FUSED_K1 else None)


class _MergingLevel(_LevelBase):
    def __init__(self, mod):
        super().__init__(mod)
        pc = mod.past_conv
        f = fold_wrapper(pc, "d")                   # 1 -> C, 1x1x1, BN, SiLU: evaluated inside the merge kernel
        self.past_w = f.w.reshape(-1)[:self.C].contiguous()
        self.past_scale, self.past_shift = f.scale[:self.C].contiguous(), f.shift[:self.C].contiguous()
        self.fusion = mod.spatial_fusion
        if self.fusion:
            self.conv5 = fold_wrapper(mod.fuse.conv_5x5, "d")
            self.fuse_conv = SepConv(mod.fuse.conv_fuse)
        self.up = ConvexUp(mod.convex_upsample)

    def merge(self, vol, samples, prev_info, resize_memory):
        """coarse.py:84-105 / fine.py:105-122 -> (cat4, sorted candidates): the merged, D-sorted volume sits in the first C
        channels of `cat4`, whose other 3C channels PyramidFusion fills (module.py:412-421)."""
        B, C, D0, H, W = vol.shape
        K = self.topk
        memory = prev_info.get('cost_memory', None)
        mem_s = mem_c = None
        if memory is not None and prev_info.get('use_past_cost', False):
            mem_s, mem_c = memory['disp_sample'], memory['cost_volume']
            if resize_memory:                                           # coarse.py:91-96
                mem_s, mem_c = resize_bilinear_pair(mem_s, mem_c, (H, W), W / mem_s.shape[-1], 1.0)
            mem_s, mem_c = _lib.contiguous(mem_s), _lib.contiguous(mem_c)
        Dm = D0 + K
        nch = 4 * C if self.fusion else C
        cat4 = torch.empty((B, nch, Dm, H, W), device=vol.device, dtype=torch.float32)
        samp = torch.empty((B, Dm, H, W), device=vol.device, dtype=torch.float32)
        x0 = cat4[:, :C]
        vb, vc = _strides5(vol); ob, oc = _strides5(x0)
        rc = _lib.lib().ts_merge_candidates_fwd(_lib.ptr(vol), _lib.ptr(samples), _lib.ptr(mem_s), _lib.ptr(mem_c),
                                                _lib.ptr(self.past_w), _lib.ptr(self.past_scale), _lib.ptr(self.past_shift),
                                                _lib.ptr(samp), _lib.ptr(x0), B, C, D0, K, H, W, vb, vc, ob, oc, _stream())
        _lib.check(rc, "ts_merge_candidates_fwd")
        return cat4, samp

    def fuse(self, cat4):
        """PyramidFusion (module.py:412-421) on the merged volume in cat4[:, :C]; without spatial fusion the volume itself."""
        C = self.C
        x0 = cat4[:, :C]
        if not self.fusion:
            return x0
        br = Branch("pool")
        with br:
            pool5(x0, cat4[:, 2 * C:3 * C], cat4[:, 3 * C:])
        conv_d(x0, self.conv5, 5, 1, 1, 2, out=cat4[:, C:2 * C])
        br.join()
        return self.fuse_conv(cat4)

    def merge_fuse_predict(self, vol, samples, prev_info, feat, resize_memory, mask=None, next_range=None):
        cat4, samp = self.merge(vol, samples, prev_info, resize_memory)
        cost, off = self.heads(self.fuse(cat4))
        disp, _, _ = TF.topk_softargmax(cost, samp, off, k=self.topk)
        if callable(mask):
            mask = mask()                    # produced on another stream: the callable joins it
        if next_range is not None:           # (range, leading candidate planes): upsample + next level's candidates, one launch
            return self.up.with_candidates(feat, disp, mask, *next_range), cost, off, samp
        return self.up(feat, disp, mask), cost, off, samp


class NativeCoarse(_MergingLevel):
    def early(self, left, right):
        """Everything of the level that does not look at the temporal state (coarse.py:79-83): cost volume + init3d."""
        return self.init3d(TF.block_cost(left, right, int(self.mod.num_sample), self.scales))

    def __call__(self, left, right, prev_info, mask=None, next_range=None, vol=None):
        if vol is None:
            vol = self.early(left, right)
        return self.merge_fuse_predict(vol, None, prev_info, left, resize_memory=True, mask=mask, next_range=next_range)


class NativeFine(_MergingLevel):
    def __init__(self, mod):
        super().__init__(mod)
        self.split_reference_half()

    def early(self, left, right, ds, terms=None):
        """Cost volume + first layer + init3d of the level (fine.py:96-103): everything before the temporal merge."""
        lt, rt = terms() if callable(terms) else (terms if terms is not None else self.feature_terms(left, right))
        if rt is not None:
            return self.init3d_from(self.first_layer_fused(left, right, ds, lt, rt))
        return self.init3d(TF.block_cost_warped(left, right, ds, self.scales), lt)

    def __call__(self, left, right, ds, prev_info, mask=None, terms=None, next_range=None, vol=None):
        """terms: feature_terms(left, right), or a callable returning it (produced on another stream: the callable joins it).
        vol: early(...) made beforehand (a pipelined pass runs it as a stage of its own)."""
        if vol is None:
            vol = self.early(left, right, ds, terms)
        return self.merge_fuse_predict(vol, ds, prev_info, left, resize_memory=False, mask=mask, next_range=next_range)


class NativePrecise(_LevelBase):
    def __init__(self, mod):
        super().__init__(mod)
        u = mod.refinement
        self.enc = [fold_wrapper(m, "hw") for m in (u.conv2[0], u.conv2[1], u.conv4[0], u.conv4[1])]
        self.enc_stride = [u.conv2[0].stride[0], u.conv2[1].stride[0], u.conv4[0].stride[0], u.conv4[1].stride[0]]
        self.fuse = [fold_wrapper(u.fuse[0], "hw"), fold_wrapper(u.fuse[1], "hw")]
        self.deconv4 = Folded(u.deconv4.weight, u.deconv4.bias, u.deconv4.norm, _act_code(u.deconv4.activation), True, "deconv2d")
        self.concat = fold_wrapper(u.concat, "hw")
        self.deconv2 = Folded(u.deconv2.weight, u.deconv2.bias, None, ACT_NONE, True, "deconv2d")
        self.in_planes = mod.in_planes
        self.split_reference_half()

    @staticmethod
    def _c2d(x, f, stride, out=None):
        return conv_hw(x, f, stride, 1, out=out)

    def encode(self, left_image, right_image, cat4, s2_left):
        """UNet.encoder (module.py:459-466) for both views without stacking them first: the left view's 1/2-resolution
        features are written where the decoder concatenates them (`s2_left`, a channel slice, module.py:488), the 1/4
        features of both views go straight into their channel slice of `cat4`.
        Batch 1: every layer runs once on the two views as ONE batch of two (the batch stride handed to the kernel is
        the distance between the two allocations).  Larger batches: the first three layers run once per view (a kernel
        takes one batch stride), which still replaces the three stacking copies by nothing."""
        B = left_image.shape[0]
        e, st = self.enc, self.enc_stride
        li, ri = left_image.unsqueeze(2), right_image.unsqueeze(2)
        s2_right = torch.empty((B,) + tuple(s2_left.shape[1:]), device=li.device, dtype=torch.float32)
        if B == 1:
            x = conv_hw(li, e[0], st[0], 1, second=ri)
            conv_hw(x[:1], e[1], st[1], 1, out=s2_left, second=x[1:], out_second=s2_right)
            x = conv_hw(s2_left, e[2], st[2], 1, second=s2_right)
        else:
            Hx, Wx = (li.shape[-2] - 1) // st[0] + 1, (li.shape[-1] - 1) // st[0] + 1
            x = torch.empty((2 * B, e[0].cout, 1, Hx, Wx), device=li.device, dtype=torch.float32)
            conv_hw(li, e[0], st[0], 1, out=x[:B]); conv_hw(ri, e[0], st[0], 1, out=x[B:])
            conv_hw(x[:B], e[1], st[1], 1, out=s2_left); conv_hw(x[B:], e[1], st[1], 1, out=s2_right)
            Hy, Wy = (s2_left.shape[-2] - 1) // st[2] + 1, (s2_left.shape[-1] - 1) // st[2] + 1
            y = torch.empty((2 * B, e[2].cout, 1, Hy, Wy), device=li.device, dtype=torch.float32)
            conv_hw(s2_left, e[2], st[2], 1, out=y[:B]); conv_hw(s2_right, e[2], st[2], 1, out=y[B:])
            x = y
        self._c2d(x, e[3], st[3], out=cat4[:, self.in_planes:].unsqueeze(2))

    def _deconv(self, x, f, out, out_bstride):
        B, Cin, H, W = x.shape
        L = _lib.lib()
        if X6 and X6S and L.ts_conv3d_hw_x6s_supported(Cin, f.cout, H, W, X6S_T4) and x6s_grid(B, f.cout, 1, H, W, X6S_T4) >= _X6S_MIN_GRID:
            rc = L.ts_conv3d_hw_x6s_fwd(_lib.ptr(x), _lib.ptr(x6s_weights(f, X6S_T4)), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                                        B, Cin, f.cout, 1, H, W, X6S_T4, f.act, 0.0, Cin * H * W, H * W, out_bstride, 4 * H * W, _stream())
            _lib.check(rc, "ts_conv3d_hw_x6s_fwd")
            return
        rc = _lib.lib().ts_deconv2d_k4s2_fwd(_lib.ptr(x), _lib.ptr(f.w), _lib.ptr(f.scale), _lib.ptr(f.shift), _lib.ptr(out),
                                             B, Cin, f.cout, H, W, f.act, out_bstride, _stream())
        _lib.check(rc, "ts_deconv2d_k4s2_fwd")

    def unet_features(self, left, right, left_image, right_image):
        """Everything of the refinement UNet that does not depend on a disparity: the image encoder
        (module.py:459-466), the [feature | spx4] concatenation both views feed to block_cost, and the
        decoder down to the 9-tap upsampling mask (module.py:484-491).  Independent of the pyramid, so
        the aggregator runs it concurrently with the coarse and fine levels."""
        both, lterm, st = self.unet_encode(left, right, left_image, right_image)
        return both, (self.unet_decode(st), lterm)

    def unet_encode(self, left, right, left_image, right_image):
        """First part of unet_features: the image encoder of both views, the [feature | spx4] concatenation and the first layer's
        per-pixel terms -- what the 1/4 level's cost volume needs.  Returns (both, lterm, state for unet_decode)."""
        B, Cf, H, W = left.shape
        both = torch.empty((2 * B, 2 * Cf, H, W), device=left.device, dtype=torch.float32)   # [left | right] x [feat | spx4]
        lcat, rcat = both[:B], both[B:]
        copy_rows(left, lcat[:, :Cf]); copy_rows(right, rcat[:, :Cf])
        C32, C2 = self.deconv4.cout, self.enc[1].cout
        cat2 = torch.empty((B, C32 + C2, 2 * H, 2 * W), device=left.device, dtype=torch.float32)      # [deconv4 | s2 of the left view]
        self.encode(left_image, right_image, both, cat2[:, C32:].unsqueeze(2))
        lterm = self.feature_terms(lcat, rcat)
        return both, lterm, (lcat, cat2, B, H, W)

    def unet_decode(self, st):
        """Second part: the decoder down to the 9-tap upsampling mask (module.py:484-491), needed by the last launch of a pass only."""
        lcat, cat2, B, H, W = st
        f = self._c2d(self._c2d(lcat.unsqueeze(2), self.fuse[0], 1), self.fuse[1], 1).squeeze(2)
        self._deconv(_lib.contiguous(f), self.deconv4, cat2, cat2.stride(0))
        g = _lib.contiguous(self._c2d(cat2.unsqueeze(2), self.concat, 1).squeeze(2))
        mask = torch.empty((B, 9, 4 * H, 4 * W), device=lcat.device, dtype=torch.float32)
        self._deconv(g, self.deconv2, mask, mask.stride(0))
        return mask

    def early(self, both, lterm, ds):
        """Cost volume + first layer + init3d of the 1/4 level (precise.py:88-93)."""
        B = both.shape[0] // 2
        lcat, rcat = both[:B], both[B:]
        lt, rt = lterm
        if rt is not None:
            return self.init3d_from(self.first_layer_fused(lcat, rcat, ds, lt, rt))
        return self.init3d(TF.block_cost_warped(lcat, rcat, ds, self.scales), lt)

    def __call__(self, both, mask_lterm, ds, prev_info, vol=None):
        B = both.shape[0] // 2
        H, W = both.shape[-2:]
        mask, lterm = mask_lterm
        mask = mask() if callable(mask) else mask
        if vol is None:
            vol = self.early(both, lterm, ds)
        cost, off = self.heads(vol)
        disp, mem_s, mem_c = TF.topk_softargmax(cost, ds, off, k=self.topk)
        full = torch.empty((B, 1, 4 * H, 4 * W), device=both.device, dtype=torch.float32)
        rc = _lib.lib().ts_unet_upsample_fwd(_lib.ptr(mask), _lib.ptr(disp), _lib.ptr(full), B, H, W, 4 * H, 4 * W, _stream())
        _lib.check(rc, "ts_unet_upsample_fwd")
        prev_info['prev_disp'] = full
        mem_s, mem_c = resize_bilinear_pair(mem_s, mem_c, (H // 2, W // 2), 0.5, 1.0)                  # precise.py:100-103
        prev_info['cost_memory'] = {'disp_sample': mem_s, 'cost_volume': mem_c}
        return full, disp, cost, off, ds


class NativeAggregator:
    """Callable with the signature and outputs of TEMPORALSTEREO.forward, eval mode, HIP kernels only."""

    def __init__(self, net, private_streams=False):
        if net.training:
            raise RuntimeError("NativeAggregator folds BatchNorm: put the module in eval() first")
        if any(p.device.type != "cuda" for p in net.parameters()):
            raise RuntimeError("NativeAggregator needs the module on the GPU (there is no CPU path)")
        # The coarse and fine levels are a chain of ~70 small kernels (grids of 50-500 workgroups) that
        # leave most of the 256 CUs idle; the disparity-independent half of the refinement UNet is
        # ~0.6 ms of wide kernels.  The chain runs on a HIGH-priority stream so that its workgroups are
        # dispatched ahead of the wide kernels' (which fill whatever is left) and both finish together.
        dev = next(net.parameters()).device
        self.device = dev
        with torch.cuda.device(dev):
            self._build(net)
            # The fold runs as ~2,000 framework launches on the CALLER's stream, and the two-phase / pipelined schedules start their
            # helper streams WITHOUT waiting for that stream (engine.begin, _staged_pass): a first pass issued right behind the build
            # could read half-folded weights -- and make its bf16-split copies from them, for good.  Seen once in nine runs of the
            # two-phase schedule on freshly built engines (round 5: a KITTI-size first frame 57 px off); the build is rare, so it
            # simply completes here.
            torch.cuda.synchronize(dev)
        self.fast, self.aux = qualified_streams(dev, 2, private=private_streams)
        # Stage streams of a pipelined pass (_staged_pass).  Rounds 2-5: three (coarse level | fine level | UNet half + 1/4-level tail on the
        # caller's stream).  Round 6: the pipeline was bound by its longest STAGE, not by the chip -- a fourth stream for the UNet half
        # (batch 4 1843 -> 1951 pairs/s), the fine level's feature-only work moved from the coarse stage (the longest) to the head of its own
        # (batch 1 1336 -> 1434-1443), each level cut at its temporal merge (cost volume + init3d | merge, fusion, heads, regression) and
        # the UNet half at its decoder.  TS_STAGES = 3 | 4 | 5 ... 8: how many streams (default 8); the last stage is always the caller's.
        self._extra_streams = qualified_streams(dev, 3, private=True)
        self._stage_sets = {}
        self.pipeline_slot = None            # set by the engine while it records one of its double-buffered plans
        self.overlap = True

    def _build(self, net):
        """Fold BatchNorm / bias into per-channel scale and shift and re-lay the weights out for the kernels.  The
        result is a private COPY of the parameters: call again (InferenceEngine.refresh does) after they change."""
        self.tape = Tape()
        with self.tape:
            self.coarse, self.fine, self.precise = NativeCoarse(net.coarse), NativeFine(net.fine), NativePrecise(net.precise)

    def refresh_weights(self):
        """The module's parameters / running statistics changed: re-make every folded array IN PLACE (Tape) -- a few launches on the
        current stream; recorded plans and captured graphs that point into the arrays stay valid.  Raises if some array of this
        model cannot be expressed by the tape (rebuild then)."""
        with torch.cuda.device(self.device):
            self.tape.refresh(self.device)

    def _coarse_level(self, l16, r16, prev_info, out, mask=None, vol=None):
        rng = 4
        disps, costs, offs, samples, ranges = out
        lm = prev_info.get('local_map', None)                       # fine.py:89-93: local-map candidates go first
        nl = lm.shape[1] if (lm is not None and prev_info.get('local_map_size', 0) > 0) else 0
        (d, low, high, ds), c, o, s = self.coarse(_lib.contiguous(l16), _lib.contiguous(r16), prev_info, mask, (rng, nl), vol=vol)
        if nl:
            resize_bilinear(lm, d.shape[-2:], d.shape[-1] / lm.shape[-1], out=ds[:, :nl])
        disps.append(d); costs.append(c); offs.append(o); samples.append(s); ranges.append({'low': low, 'high': high})
        return ds

    def _fine_level(self, l8, r8, ds, prev_info, out, mask=None, terms=None, vol=None):
        rng = 4
        disps, costs, offs, samples, ranges = out
        (d, low, high, ds), c, o, s = self.fine(_lib.contiguous(l8), _lib.contiguous(r8), ds, prev_info, mask, terms, (rng, 0), vol=vol)
        disps.append(d); costs.append(c); offs.append(o); samples.append(s); ranges.append({'low': low, 'high': high})
        return ds

    def _stages_for(self, batch):
        """role -> stream (None: the caller's) of a pipelined pass.  Batch 1: six streams, the levels cut at their temporal merge (more
        than six share hardware queues and the pipeline collapses: 1512 -> 860 pairs/s at seven; `tail` on a stream of its own loses in
        every combination).  From batch 2 on the launches fill the chip and the finer cut of the levels only adds edges (1900 vs 1960 at
        batch 4): five streams, the UNet half -- the longest stage -- cut at its decoder (1983 -> 1991 at batch 4, 2017 -> 2058 at 8).
        TS_STAGES=3..6 and TS_STAGE_ORDER=role,role,... override (A/B runs)."""
        nst = int(os.environ.get("TS_STAGES", "6" if batch == 1 else "5"))
        key = (nst, os.environ.get("TS_STAGE_ORDER", ""), batch == 1)
        smap = os.environ.get("TS_STAGE_MAP", "")
        if smap:
            # experiment: explicit role -> stream, e.g. "unet:2,coarse2:3,fine2:4,unet2:4,tail:m" (0 fast, 1 aux, 2-4 extras, m caller's)
            key = ("map", smap)
            if key not in self._stage_sets:
                pool = [self.fast, self.aux] + list(self._extra_streams)
                S = {"coarse": self.fast, "fine": self.aux, "unet": None, "coarse2": self.fast, "fine2": self.aux, "tail": None, "unet2": None}
                for item in smap.split(","):
                    role, idx = item.split(":")
                    S[role] = None if idx == "m" else pool[int(idx)]
                self._stage_sets[key] = S
            return self._stage_sets[key]
        if key not in self._stage_sets:
            extra = self._extra_streams[:max(0, min(nst, 6) - 3)]
            S = {"coarse": self.fast, "fine": self.aux}
            order = [r for r in os.environ.get("TS_STAGE_ORDER", "unet,coarse2,fine2" if batch == 1 else "unet,unet2").split(",") if r]
            order += [r for r in ("unet", "coarse2", "fine2", "tail", "unet2") if r not in order]
            parent = {"unet": None, "coarse2": "coarse", "fine2": "fine", "tail": None, "unet2": "unet"}
            for i, role in enumerate(order):
                S[role] = extra[i] if i < len(extra) else None
            for role in order:                       # roles without a stream of their own: their predecessor's (or the caller's)
                if S[role] is None and parent[role]:
                    S[role] = S[parent[role]]
            self._stage_sets[key] = S
        return self._stage_sets[key]

    def _staged_pass(self, l4, l8, l16, r4, r8, r16, left_image, right_image, prev_info, out):
        """Several passes in flight (engine.py, pipeline >= 2): the pass as a pipeline of stages, one per stream -- `fast`:
        upsampling logits + coarse level, `aux`: fine level, `wide` (round 6): the UNet half that needs no disparity, caller's
        stream: the 1/4-level tail -- with three cross-stream edges (coarse -> fine, fine -> tail, wide -> tail) and no branches inside a
        stage: every edge towards a stream that is still busy with the previous pass's stage would stall this pass
        behind it (measured: 773 pairs/s with the latency-mode branches kept, 882 without, depth 2).  None of the
        engine's streams waits for the caller's stream, where the tail of the previous pass is still running on other
        buffers; they wait for the pass that last used THESE buffers (named event, recorded at the end)."""
        L = _lib.lib()
        main = torch.cuda.current_stream()
        slot = self.pipeline_slot
        P = lambda st: _lib.ctypes.c_void_p(st.cuda_stream)
        # by role; a role without a stream of its own shares its predecessor's, or runs on the caller's stream
        S = {k: (v if v is not None else main) for k, v in self._stages_for(l4.shape[0]).items()}
        for st in {id(v): v for v in S.values() if v is not main}.values():
            _lib.check(L.ts_event_wait(slot, P(st)), "ts_event_wait")          # the pass that last used THESE buffers is done
        on = torch.cuda.stream

        def edge(a, b):
            if a is not b:
                _edge(a, b)
        try:
            _chunk_cap(8)
            l16c, r16c, l8c, r8c = _lib.contiguous(l16), _lib.contiguous(r16), _lib.contiguous(l8), _lib.contiguous(r8)
            with on(S["coarse"]):
                mc = self.coarse.up.mask(l16c)
                cvol = self.coarse.early(l16c, r16c)
            with on(S["fine"]):                        # feature-only work of the fine level at the head of ITS stage
                mf = self.fine.up.mask(l8c)
                ltf = self.fine.feature_terms(l8c, r8c)
            with on(S["unet"]):
                both, lterm, ust = self.precise.unet_encode(l4, r4, left_image, right_image)
            edge(S["unet"], S["unet2"])
            with on(S["unet2"]):
                mask = self.precise.unet_decode(ust)
            edge(S["coarse"], S["coarse2"])
            with on(S["coarse2"]):
                ds = self._coarse_level(l16, r16, prev_info, out, lambda: mc, vol=cvol)
            edge(S["coarse2"], S["fine"])
            with on(S["fine"]):
                fvol = self.fine.early(l8c, r8c, ds, ltf)
            edge(S["fine"], S["fine2"])
            with on(S["fine2"]):
                ds = self._fine_level(l8, r8, ds, prev_info, out, lambda: mf, None, vol=fvol)
            edge(S["fine2"], S["tail"]); edge(S["unet"], S["tail"])
            with on(S["tail"]):
                pvol = self.precise.early(both, lterm, ds)
            edge(S["tail"], main); edge(S["unet2"], main)
            res = self.precise(both, (mask, lterm), ds, prev_info, vol=pvol)
            _lib.check(L.ts_event_record(slot, P(main)), "ts_event_record")
        finally:
            _chunk_cap(32)
        return res

    # ---------------------------------------------------------------------------------------------- two-phase pass
    # A temporal sequence has a true dependency from frame to frame -- but only from the candidate merge of the coarse level
    # onwards (coarse.py:84-105: the cost memory enters there; the local map at the fine level's candidates, fine.py:89-93).
    # Everything before it looks at the NEW frame only: the coarse cost volume + init3d, the upsampling logits, the fine
    # level's left term and the whole disparity-independent half of the refinement UNet (~40 % of a pass's kernel time).
    # `begin` issues that part on the two helper streams, `finish` the rest once the state is there; between the two the
    # caller runs update_map for the previous frame.  On the device, begin(t+1) overlaps the 1/4-level tail of frame t and its
    # update_map, which run on the caller's stream.  The engine alternates two buffer sets so that begin(t+1) never writes
    # what finish(t) still reads; `slots` = (set-is-free event, logits-are-ready event).
    @torch.no_grad()
    def begin(self, left_feats, right_feats, left_image, right_image, slots):
        with torch.cuda.device(self.device):
            L = _lib.lib()
            P = lambda st: _lib.ctypes.c_void_p(st.cuda_stream)
            l4, l8, l16 = left_feats
            r4, r8, r16 = right_feats
            left_image, right_image = _lib.contiguous(left_image), _lib.contiguous(right_image)
            _lib.check(L.ts_event_wait(slots[0], P(self.fast)), "ts_event_wait")       # the pass that last used this set is done
            _lib.check(L.ts_event_wait(slots[0], P(self.aux)), "ts_event_wait")
            try:
                _chunk_cap(8)
                with torch.cuda.stream(self.aux):
                    mc, mf = self.coarse.up.mask(_lib.contiguous(l16)), self.fine.up.mask(_lib.contiguous(l8))
                    ltf = self.fine.feature_terms(_lib.contiguous(l8), _lib.contiguous(r8))
                    _lib.check(L.ts_event_record(slots[1], P(self.aux)), "ts_event_record")
                with torch.cuda.stream(self.fast):
                    vol = self.coarse.early(_lib.contiguous(l16), _lib.contiguous(r16))
                with torch.cuda.stream(self.aux):
                    both, mask = self.precise.unet_features(l4, r4, left_image, right_image)
            finally:
                _chunk_cap(32)
            return dict(mc=mc, mf=mf, ltf=ltf, vol=vol, both=both, mask=mask, feats=(l8, l16, r8, r16), slots=slots)

    @torch.no_grad()
    def finish(self, ctx, prev_info):
        with torch.cuda.device(self.device):
            L = _lib.lib()
            P = lambda st: _lib.ctypes.c_void_p(st.cuda_stream)
            main = torch.cuda.current_stream()
            l8, l16, r8, r16 = ctx["feats"]
            out = ([], [], [], [], [])
            disps, costs, offs, samples, ranges = out
            _edge(main, self.fast)                       # the state (update_map, copies) was produced on the caller's stream
            try:
                _chunk_cap(8)
                with torch.cuda.stream(self.fast):
                    _lib.check(L.ts_event_wait(ctx["slots"][1], P(self.fast)), "ts_event_wait")     # logits + left term (not the UNet half)
                    ds = self._coarse_level(l16, r16, prev_info, out, ctx["mc"], vol=ctx["vol"])
                    ds = self._fine_level(l8, r8, ds, prev_info, out, ctx["mf"], ctx["ltf"])
                _edge(self.fast, main)
                _edge(self.aux, main)                    # the UNet half of begin()
                full, d, c, o, s = self.precise(ctx["both"], ctx["mask"], ds, prev_info)
                _lib.check(L.ts_event_record(ctx["slots"][0], P(main)), "ts_event_record")
            finally:
                _chunk_cap(32)
            disps += [d, full]; costs.append(c); offs.append(o); samples.append(s)
            return disps[::-1], costs[::-1], samples[::-1], offs[::-1], ranges[::-1], prev_info

    @torch.no_grad()
    def __call__(self, left_feats, right_feats, left_image, right_image, prev_info):
        if left_image.device != self.device:
            raise RuntimeError("NativeAggregator lives on %s, inputs are on %s" % (self.device, left_image.device))
        with torch.cuda.device(self.device):      # the launch stream is the CURRENT device's current stream
            return self._pass(left_feats, right_feats, left_image, right_image, prev_info)

    def _pass(self, left_feats, right_feats, left_image, right_image, prev_info):
        l4, l8, l16 = left_feats
        r4, r8, r16 = right_feats
        out = ([], [], [], [], [])
        disps, costs, offs, samples, ranges = out
        left_image, right_image = _lib.contiguous(left_image), _lib.contiguous(right_image)
        if self.overlap and self.pipeline_slot is not None:
            full, d, c, o, s = self._staged_pass(l4, l8, l16, r4, r8, r16, left_image, right_image, prev_info, out)
        elif self.overlap:
            # Every use of `fast` starts by waiting on an event of the caller's stream and ends with the
            # caller's stream waiting on it, so tensors allocated under it are safe to hand over.
            main = torch.cuda.current_stream()
            mainp, fastp = _lib.ctypes.c_void_p(main.cuda_stream), _lib.ctypes.c_void_p(self.fast.cuda_stream)
            aux = self.aux
            _lib.check(_lib.lib().ts_stream_fork(mainp, fastp), "ts_stream_fork")
            _edge(main, aux)
            _PAR["on"], _PAR["aux"] = True, aux
            try:
                # Short K chunks (small LDS tiles) while three streams share the CUs: measured 1.53 ms/pair
                # with cap 8 everywhere vs 1.58 with the residency heuristic's 16/32 (a 65-106 KB tile of a
                # chain kernel cannot be placed next to the wide kernels' workgroups and waits for a CU to drain).
                _chunk_cap(8)
                # convex-upsampling logits of the coarse and fine levels depend on the features only
                with torch.cuda.stream(aux):
                    mc, mf = self.coarse.up.mask(_lib.contiguous(l16)), self.fine.up.mask(_lib.contiguous(l8))
                    ltf = self.fine.feature_terms(_lib.contiguous(l8), _lib.contiguous(r8))

                waited = []

                def joined(m):
                    def get():
                        if not waited:       # one edge covers all three: `aux` is in-order and they were enqueued together
                            _edge(aux, torch.cuda.current_stream())
                            waited.append(True)
                        return m
                    return get
                # Issue order = what the host reaches first: the chain is the critical path, so its coarse
                # level goes out before the ten wide launches (they have ~1 ms of slack), the fine level after.
                with torch.cuda.stream(self.fast):
                    ds = self._coarse_level(l16, r16, prev_info, out, joined(mc))
                both, mask = self.precise.unet_features(l4, r4, left_image, right_image)
                with torch.cuda.stream(self.fast):
                    ds = self._fine_level(l8, r8, ds, prev_info, out, joined(mf), joined(ltf))
                _lib.check(_lib.lib().ts_stream_fork(fastp, mainp), "ts_stream_fork")
                full, d, c, o, s = self.precise(both, mask, ds, prev_info)
            finally:
                _PAR["on"], _PAR["aux"] = False, None
                _chunk_cap(32)
        else:
            ds = self._coarse_level(l16, r16, prev_info, out)
            ds = self._fine_level(l8, r8, ds, prev_info, out)
            both, mask = self.precise.unet_features(l4, r4, left_image, right_image)
            full, d, c, o, s = self.precise(both, mask, ds, prev_info)
        disps += [d, full]; costs.append(c); offs.append(o); samples.append(s)
        return disps[::-1], costs[::-1], samples[::-1], offs[::-1], ranges[::-1], prev_info
