"""Full-size parity harness shared by tests/test_fullsize_gpu.py and tools/parity_report.py (not collected).

BASELINE.json's configurations at their STATED batches, as sequences: every frame is one oracle pass (CPU, `trace` of
every stage boundary) and one pass of the product path, with the temporal state update (`update_map`) between frames
on both sides.  Three ways of comparing:

  per-op teacher forcing     each HIP stage is fed the ORACLE's stage inputs: errors cannot compound, discrete steps
                             (top-k, sort) decide on identical bits
  per-level teacher forcing  each pyramid level is fed the oracle's level inputs: the only discrete decisions on OUR
                             numbers are the level's own top-k / sort, so a pixel may differ only where the oracle's
                             own decision was a near-tie (margin below twice the measured cost error)
  end to end                 nothing forced: |dEPE| against the oracle with every pixel included
"""
import json
import os

import numpy as np
import torch

import synth

# name -> geometry.  `frames` > 1: temporal sequence (FRAME_IDXS), `n_local` = LOCAL_MAP_SIZE
CONFIGS = {
    "configs[1] things 544x960 D=192 single B=1": dict(H=544, W=960, num_sample=12, B=1, frames=1, n_local=0, fx=1050.0 * 544 / 540, baseline=1.0),
    "configs[2] things 544x960 D=192 T=2 B=4": dict(H=544, W=960, num_sample=12, B=4, frames=2, n_local=1, fx=1050.0 * 544 / 540, baseline=1.0),
    "configs[3] tartanair 480x640 D=128 seq B=8": dict(H=480, W=640, num_sample=8, B=8, frames=4, n_local=3, fx=320.0, baseline=0.25),
    "configs[4] kitti 384x1248 D=192 temporal B=2": dict(H=384, W=1248, num_sample=12, B=2, frames=2, n_local=3, fx=721.5377, baseline=0.54),
}


def intrinsics(c):
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = c["fx"]
    K[0, 2], K[1, 2] = c["W"] / 2 - 0.5, c["H"] / 2 - 0.5
    return np.broadcast_to(K, (c["B"], 4, 4)).copy()


def to_dev(obj, dev, dtype=None):
    if torch.is_tensor(obj):
        obj = obj.to(dev)
        return obj.to(dtype) if (dtype is not None and obj.is_floating_point()) else obj
    if isinstance(obj, dict):
        return {k: to_dev(v, dev, dtype) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_dev(v, dev, dtype) for v in obj)
    return obj


class Case:
    """One configuration + seed: the calibrated synthetic network, the per-frame inputs and poses."""

    def __init__(self, c, seed, dev):
        import bench
        self.c, self.seed, self.dev = c, seed, dev
        self.net = bench.build_model(dev, seed, c["num_sample"])
        self.frames_cpu = [bench.make_inputs(torch.device("cpu"), seed + 1000 * t, c["B"], (c["H"], c["W"])) for t in range(c["frames"])]
        self.frames_gpu = [to_dev(f, dev) for f in self.frames_cpu]
        self.K = torch.from_numpy(intrinsics(c))
        self.T = [torch.from_numpy(synth.small_motion(seed + t, c["B"])) for t in range(c["frames"])]
        self.eye = torch.eye(4).expand(c["B"], 4, 4).contiguous()
        bench.calibrate_batchnorm(self.net, self.frames_gpu[0])
        if c["frames"] > 1:
            # A temporal model's BatchNorm statistics come from temporal frames: calibrated on frame 0 alone, `past_conv`
            # (1 -> C on the cost memory) would have seen only the zero memory of single-frame mode (variance ~ 0) and
            # would amplify a real memory by 1/sqrt(eps) ~ 316 -- cost logits in the thousands, nothing like a trained
            # network.  So: frame 0 -> update_map -> calibrate on frame 1 WITH its temporal state.
            # (the state update of the calibration runs through the ORACLE on the CPU: the product's fused update accumulates its splat
            # with fp32 atomics, whose order -- and with it the calibrated statistics, i.e. the WEIGHTS both sides are then compared on,
            # and which near-ties the network happens to have -- differed from run to run: one in five full-suite runs tripped a bar)
            from oracle import temporal as otemp
            with torch.no_grad():
                info = self.net.eval()(*self.frames_gpu[0], {})[5]
            info = to_dev({k: v for k, v in info.items() if k in ("prev_disp", "cost_memory")}, "cpu")
            info = otemp.update_map(info, self.K, self.T[1], self.eye, c["baseline"], c["H"], c["W"], use_past_cost=True, local_map_size=c["n_local"])
            bench.calibrate_batchnorm(self.net, self.frames_gpu[1], to_dev(state_for_aggregation(info), dev))
        self.sd = {k: v.detach().cpu() for k, v in self.net.state_dict().items()}
        self.cfg = dict(coarse=dict(num_sample=c["num_sample"]))
        self.max_disp = 16 * c["num_sample"]

    # ------------------------------------------------------------------ oracle side
    def oracle_frame(self, t, prev, dtype=torch.float32):
        """-> (outputs, trace, prev_info as handed to the aggregation)."""
        from oracle import aggregation as oagg
        from oracle import temporal as otemp
        c = self.c
        cast = lambda o: to_dev(o, "cpu", dtype)
        prev = dict(prev)
        if t > 0:
            prev = otemp.update_map(prev, cast(self.K), cast(self.T[t]), cast(self.eye), c["baseline"], c["H"], c["W"],
                                    use_past_cost=True, local_map_size=c["n_local"])
        trace = {}
        lf, rf, il, ir = cast(self.frames_cpu[t])
        with torch.no_grad():
            out = oagg.aggregate(cast(self.sd), lf, rf, il, ir, dict(prev), cfg=self.cfg, trace=trace)
        return out, trace, prev

    # ------------------------------------------------------------------ product side
    def native_update(self, t, prev):
        from temporalstereo_amd import temporal
        c = self.c
        dev = self.dev
        return temporal.update_map(dict(prev), self.K.to(dev), self.T[t].to(dev), self.eye.to(dev), c["baseline"], c["H"], c["W"],
                                   use_past_cost=True, local_map_size=c["n_local"])


CKPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_planted.npz")


def load_checkpoint(path=None):
    """The committed contractive checkpoint (tools/train_checkpoint.py: this repository's own training step on planted scenes)."""
    with np.load(path or os.environ.get("TS_CKPT", CKPT)) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


class PlantedCase(Case):
    """One configuration + seed on a planted-disparity scene (synth.stereo_sequence) with the trained checkpoint: inputs that are
    stereo, poses that move the scene rigidly, a ground truth per frame, weights that make the pyramid contractive."""

    def __init__(self, c, seed, dev, ckpt=None):
        import bench
        self.c, self.seed, self.dev = c, seed, dev
        self.max_disp = 16 * c["num_sample"]
        self.net = bench.build_model(dev, seed, c["num_sample"])
        self.net.load_state_dict(load_checkpoint(ckpt), strict=True)
        self.net.eval()
        sc = synth.stereo_sequence(seed, c["B"], c["H"], c["W"], frames=c["frames"], max_disp=self.max_disp, fx=c["fx"], baseline=c["baseline"])
        T = torch.from_numpy
        self.frames_cpu = [([T(x) for x in lf], [T(x) for x in rf], T(il), T(ir)) for lf, rf, il, ir in sc["frames"]]
        self.frames_gpu = [to_dev(f, dev) for f in self.frames_cpu]
        self.gt = [T(g) for g in sc["gt"]]
        self.K = T(sc["K"])
        self.T = [T(t) for t in sc["T"]]
        self.eye = torch.eye(4).expand(c["B"], 4, 4).contiguous()
        self.sd = {k: v.detach().cpu() for k, v in self.net.state_dict().items()}
        self.cfg = dict(coarse=dict(num_sample=c["num_sample"]))
        self.input_checksum = [synth.checksum([f[0], f[1], f[2], f[3]]) for f in sc["frames"]]


def epe(disp, gt, max_disp):
    """EPE = mean |d - gt| over the valid mask 0 < gt < max_disp (data/evaluation/pixel_error.py:33-63), in float64."""
    a, g = disp.detach().double().cpu(), gt.double().cpu()
    m = (g > 0) & (g < max_disp)
    return float((a - g).abs()[m].mean())


def state_for_aggregation(info):
    """The entries of prev_info the aggregation reads."""
    return {k: v for k, v in info.items() if k in ("cost_memory", "use_past_cost", "local_map", "local_map_size") and v is not None}


def delta_epe(ours, ref, seed, max_disp):
    """|EPE(ours, gt*) - EPE(ref, gt*)| with gt* = ref + N(0,1) clipped to (0, max_disp) (SURVEY.md section 8(d);
    EPE = mean |d - gt|, data/evaluation/pixel_error.py:33-63), and mean |ours - ref|."""
    a, b = ours.detach().double().cpu(), ref.detach().double().cpu()
    gen = torch.Generator().manual_seed(int(seed))
    gt = (b + torch.randn(b.shape, generator=gen, dtype=torch.float64)).clamp(0, max_disp)
    return abs(float((a - gt).abs().mean()) - float((b - gt).abs().mean())), float((a - b).abs().mean())


def top_margin(cost, k=2):
    """Per pixel: gap between the k-th and (k+1)-th best cost -- the margin of the top-k selection."""
    top = torch.topk(cost.double(), k + 1, dim=1).values
    return (top[:, k - 1] - top[:, k]).abs()


class Report:
    """Collects the measured differences; `dump` writes them next to the profiles when the directory is writable."""

    def __init__(self):
        self.rows = []

    def add(self, **kw):
        self.rows.append(kw)

    def dump(self, name):
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        try:
            os.makedirs(root, exist_ok=True)
            path = os.path.join(root, name)
            old = []
            if os.path.exists(path):
                with open(path) as fh:
                    old = json.load(fh)
            with open(path, "w") as fh:
                json.dump(old + self.rows, fh, indent=1)
        except OSError:
            pass


def compare(report, what, ours, ref, atol, rtol=0.0, **ctx):
    """max |ours - ref| <= atol + rtol * |ref| elementwise; the measured figures go to the report either way."""
    a, b = ours.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    d = (a - b).abs()
    worst = float((d - rtol * b.abs()).max())
    report.add(what=what, max_abs=float(d.max()), mean_abs=float(d.mean()), ref_max=float(b.abs().max()), atol=atol, rtol=rtol, **ctx)
    assert worst <= atol, "%s: max |diff| %.3g (ref scale %.3g) exceeds atol %.1e + rtol %.1e" % (what, float(d.max()), float(b.abs().max()), atol, rtol)
