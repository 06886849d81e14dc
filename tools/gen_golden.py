#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference).

Run in the build container only:   PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
The reference source never leaves /root/reference; only inputs / expected outputs are saved.
Inputs that are cheap to regenerate are stored as seeds (tests/synth.py) instead of bytes.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_import  # noqa: E402
import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
T = torch.from_numpy


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote %-38s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------
def functional_cases(M):
    from architecture.modeling.aggregation.utils.block_cost import block_cost
    from architecture.modeling.layers import inverse_warp_3d, project_to_3d
    from architecture.modeling.prediction.soft_argmin import SOFTARGMIN
    from architecture.modeling.prediction.argmin import ARGMIN
    from architecture.modeling.aggregation.TemporalStereo.coarse import CoarseAggregation

    seed = synth.SEED0
    # K1a integer path ---------------------------------------------------------------------
    for i, (B, C, H, W, D) in enumerate([(2, 16, 10, 14, 3), (1, 8, 8, 12, 5), (1, 8, 9, 13, 4)]):
        L = synth.normal(seed + i, "bcL", (B, C, H, W))
        R = synth.normal(seed + i, "bcR", (B, C, H, W))
        out = block_cost(T(L), T(R), D, block_cost_scale=3)
        save("block_cost_int_%d" % i, left=L, right=R, num_disp=D, scales=3, out=out)
    # K1b sampled path ---------------------------------------------------------------------
    for i, (B, C, H, W, D) in enumerate([(2, 16, 10, 14, 5), (1, 8, 8, 12, 3), (1, 8, 9, 13, 6)]):
        L = synth.normal(seed + 10 + i, "bcL", (B, C, H, W))
        R = synth.normal(seed + 10 + i, "bcR", (B, C, H, W))
        disp = synth.uniform(seed + 10 + i, "bcD", (B, D, H, W), -3.0, W + 2.0)
        disp[:, 0] = np.round(disp[:, 0])            # exact-integer candidates
        disp[:, 1, :, : W // 2] = 0.0                # zero shift
        out = block_cost(T(L), T(R), T(disp), block_cost_scale=3)
        save("block_cost_sampled_%d" % i, left=L, right=R, disp=disp, scales=3, out=out)
    # scale count 1 and 2
    L = synth.normal(seed + 20, "bcL", (1, 8, 8, 12)); R = synth.normal(seed + 20, "bcR", (1, 8, 8, 12))
    disp = synth.uniform(seed + 20, "bcD", (1, 4, 8, 12), 0.0, 6.0)
    for sc in (1, 2):
        save("block_cost_sampled_scale%d" % sc, left=L, right=R, disp=disp, scales=sc,
             out=block_cost(T(L), T(R), T(disp), block_cost_scale=sc))
        save("block_cost_int_scale%d" % sc, left=L, right=R, num_disp=4, scales=sc,
             out=block_cost(T(L), T(R), 4, block_cost_scale=sc))
    # the reference-authored 3x4 "value test" inputs (cat_fms.py:48-57 / dif_fms.py:56-65)
    H, W = 3, 4
    left = torch.linspace(1, H * W, H * W).reshape(1, 1, H, W)
    right = torch.linspace(H * W + 1, H * W * 2, H * W).reshape(1, 1, H, W)
    ds = torch.linspace(-2, 2, 5).repeat(1, H, W, 1).permute(0, 3, 1, 2).contiguous()
    save("value_test_warp", left=left, right=right, disp=ds, warped=inverse_warp_3d(right, -ds, padding_mode='zeros'))
    # K2a geometry --------------------------------------------------------------------------
    B, C, H, W = 2, 3, 6, 8
    depth = synth.uniform(seed + 30, "depth", (B, C, H, W), 1.0, 20.0)
    K = synth.sceneflow_intrinsics(B, 48, 64)
    K[:, 0] /= 8.0; K[:, 1] /= 8.0
    Tm = synth.small_motion(seed + 30, B)
    o = project_to_3d(T(depth), T(K), torch.inverse(T(K)), T(Tm))
    save("project_to_3d", depth=depth, K=K, T=Tm, triangular_depth=o['triangular_depth'],
         optical_flow=o['optical_flow'], flow_mask=o['flow_mask'], src_pixel_coord=o['src_pixel_coord'])
    o3 = project_to_3d(T(depth), T(K[:, :3, :3].copy()), None, T(Tm))
    save("project_to_3d_k3", depth=depth, K=K[:, :3, :3], T=Tm, triangular_depth=o3['triangular_depth'],
         optical_flow=o3['optical_flow'])
    # K4 regression -------------------------------------------------------------------------
    B, D, H, W = 2, 14, 6, 9
    cost = synth.normal(seed + 40, "cost", (B, D, H, W))
    samp = np.sort(synth.uniform(seed + 40, "samp", (B, D, H, W), 0, 30), axis=1)
    off = synth.uniform(seed + 40, "off", (B, D, H, W), -1, 1)
    tiny = CoarseAggregation(in_planes=8, C=8, num_sample=2)
    d, td, tc = tiny.predict_disp(T(cost), T(samp), T(off), k=2)
    save("topk_softargmax", cost=cost, samp=samp, off=off, disp=d, topk_disp=td, topk_cost=tc)
    d3, td3, tc3 = tiny.predict_disp(T(cost), T(samp), T(off), k=3)
    save("topk_softargmax_k3", cost=cost, samp=samp, off=off, disp=d3, topk_disp=td3, topk_cost=tc3)
    Dd = 48
    cost = synth.normal(seed + 41, "cost", (1, Dd, 5, 7), 3.0)
    samp = np.broadcast_to(np.arange(Dd, dtype=np.float32).reshape(1, Dd, 1, 1), (1, Dd, 5, 7)).copy()
    save("soft_argmin", cost=cost, samp=samp,
         disp=SOFTARGMIN(temperature=1.0, normalize=True)(T(cost), T(samp)),
         disp_t2=SOFTARGMIN(temperature=2.0, normalize=True)(T(cost), T(samp)),
         disp_argmin=ARGMIN(dim=1)(T(cost), T(samp)))


# ----------------------------------------------------------------------------------------------
def sibling_cases(M):
    """Dense siblings of block_cost: cat_fms / dif_fms."""
    from architecture.modeling.aggregation.utils.cat_fms import cat_fms
    from architecture.modeling.aggregation.utils.dif_fms import dif_fms
    seed = synth.SEED0 + 60
    # the reference-authored 3x4 value test (cat_fms.py:48-69, dif_fms.py:56-76): shifts -2..2
    H, W = 3, 4
    left = torch.linspace(1, H * W, H * W).reshape(1, 1, H, W)
    right = torch.linspace(H * W + 1, H * W * 2, H * W).reshape(1, 1, H, W)
    ds = torch.linspace(-2, 2, 5).repeat(1, H, W, 1).permute(0, 3, 1, 2).contiguous()
    save("fms_value_test", left=left, right=right, disp=ds, cat=cat_fms(left, right, ds), dif=dif_fms(left, right, ds))
    for i, (B, C, H, W, D, lo, hi) in enumerate([(2, 16, 10, 14, 6, -3.0, 16.0), (1, 8, 9, 13, 4, 0.0, 6.0), (1, 24, 6, 20, 9, -1.0, 21.0)]):
        L = synth.normal(seed + i, "fmL", (B, C, H, W))
        R = synth.normal(seed + i, "fmR", (B, C, H, W))
        if i == 1:
            L, R = np.abs(L), np.abs(R)                # post-ReLU style features: the (target > 0) mask is then the in-frame mask
        disp = synth.uniform(seed + i, "fmD", (B, D, H, W), lo, hi)
        disp[:, 0] = np.round(disp[:, 0])
        if i == 2:                                   # the dense use: integer candidates 0..D-1
            disp = np.broadcast_to(np.arange(D, dtype=np.float32).reshape(1, D, 1, 1), (B, D, H, W)).copy()
        save("fms_%d" % i, left=L, right=R, disp=disp, cat=cat_fms(T(L), T(R), T(disp)), dif=dif_fms(T(L), T(R), T(disp)))


# ----------------------------------------------------------------------------------------------
def build_reference_aggregator(dims):
    from architecture.modeling.aggregation.TemporalStereo.TemporalStereo import TEMPORALSTEREO
    from architecture.modeling.aggregation.TemporalStereo.coarse import CoarseAggregation
    from architecture.modeling.aggregation.TemporalStereo.fine import FineAggregation
    from architecture.modeling.aggregation.TemporalStereo.precise import PreciseAggregation
    c, f, p = dims['coarse'], dims['fine'], dims['precise']
    net = TEMPORALSTEREO(
        coarse=CoarseAggregation(in_planes=c['in_planes'], C=c['C'], num_sample=c['num_sample'], topk=2),
        fine=FineAggregation(in_planes=f['in_planes'], C=f['C'], num_sample=5, topk=2),
        precise=PreciseAggregation(in_planes=p['in_planes'], C=p['C'], num_sample=5, topk=2),
    )
    return net


def load_synth_weights(net, seed):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    import json
    tag = "x".join(str(v.shape[0]) for k, v in net.state_dict().items() if k.endswith("init3d.0.conv.0.weight"))
    with open(os.path.join(OUT, "state_shapes_%s.json" % tag), "w") as fh:
        json.dump({k: list(v) for k, v in shapes.items()}, fh, indent=0, sort_keys=True)
    vals = synth.state_values(shapes, seed)
    net.load_state_dict({k: T(v) for k, v in vals.items()}, strict=True)
    return shapes


class Recorder:
    """Records block_cost outputs and the permutations torch.sort produced inside the reference."""

    def __init__(self):
        self.raw, self.orders = [], []

    def __enter__(self):
        import architecture.modeling.aggregation.TemporalStereo.coarse as c
        import architecture.modeling.aggregation.TemporalStereo.fine as f
        import architecture.modeling.aggregation.TemporalStereo.precise as p
        self.mods = (c, f, p)
        self.orig_bc = c.block_cost
        self.orig_sort = torch.sort

        def bc(*a, **k):
            out = self.orig_bc(*a, **k)
            self.raw.append(out)
            return out

        def sort(x, *a, **k):
            r = self.orig_sort(x, *a, **k)
            self.orders.append(r[1])
            return r
        for m in self.mods:
            m.block_cost = bc
        torch.sort = sort
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m.block_cost = self.orig_bc
        torch.sort = self.orig_sort


TINY = dict(coarse=dict(in_planes=32, C=8, num_sample=4), fine=dict(in_planes=16, C=8),
            precise=dict(in_planes=8, C=8))
SCENEFLOW = dict(coarse=dict(in_planes=256, C=32, num_sample=12), fine=dict(in_planes=128, C=16),
                 precise=dict(in_planes=64, C=8))


def synthetic_prev_info(seed, B, H, W, local_maps=2):
    """A plausible temporal state at 1/8 resolution (what update_map would hand over)."""
    h, w = H // 8, W // 8
    base = synth.uniform(seed, "mem_base", (B, 1, h, w), 1.0, 6.0)
    ds = np.concatenate([base + 0.3, base - 0.4], axis=1).astype(np.float32)
    cv = synth.normal(seed, "mem_cost", (B, 2, h, w))
    lm = np.concatenate([base * 1.02 + 0.1 * k for k in range(local_maps)], axis=1).astype(np.float32)
    return ds, cv, lm


def aggregator_case(name, dims, seed, B, H, W, temporal, store_inputs, training=False):
    net = build_reference_aggregator(dims)
    load_synth_weights(net, seed)
    chans = (dims['precise']['in_planes'], dims['fine']['in_planes'], dims['coarse']['in_planes'])
    lf, rf = synth.feature_pyramid(seed, B, H, W, chans=chans)
    il, ir = synth.images(seed, B, H, W)
    prev = {}
    extra = {}
    if temporal:
        ds, cv, lm = synthetic_prev_info(seed, B, H, W)
        prev = {'cost_memory': {'disp_sample': T(ds), 'cost_volume': T(cv)}, 'use_past_cost': True,
                'local_map': T(lm), 'local_map_size': lm.shape[1]}
        extra.update(mem_disp_sample=ds, mem_cost_volume=cv, local_map=lm)
    bn_stats = {}
    if not training:
        # give the synthetic network BatchNorm statistics that match its activations (as a trained
        # network has): one calibration pass in train mode with momentum 1, then eval.
        bns = [m for m in net.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d))]
        for m in bns:
            m.momentum = 1.0
        net.train(True)
        with torch.no_grad():
            net([T(x) for x in lf], [T(x) for x in rf], T(il), T(ir), dict(prev))
        for m in bns:
            m.momentum = 0.1
        bn_stats = {"bn::" + k: v.clone() for k, v in net.state_dict().items()
                    if k.endswith("running_mean") or k.endswith("running_var")}
    net.train(training)
    with torch.no_grad(), Recorder() as rec:
        disps, costs, samples, offs, ranges, info = net([T(x) for x in lf], [T(x) for x in rf], T(il), T(ir), prev)
    arrs = dict(seed=seed, B=B, H=H, W=W, temporal=int(temporal), training=int(training),
                dims=np.array([dims['coarse']['in_planes'], dims['coarse']['C'], dims['coarse']['num_sample'],
                               dims['fine']['in_planes'], dims['fine']['C'],
                               dims['precise']['in_planes'], dims['precise']['C']]))
    arrs.update(extra)
    arrs.update(bn_stats)
    if store_inputs:     # cross-check that synth regenerates the same bytes on the test side
        arrs.update(l16_probe=lf[2][:, :2], r16_probe=rf[2][:, :2])
        for i, nm in enumerate(("full", "precise", "fine_up", "coarse_up")):
            arrs["disp_" + nm] = disps[i]
        for i, nm in enumerate(("precise", "fine", "coarse")):
            arrs["cost_" + nm] = costs[i]
            arrs["samp_" + nm] = samples[i]
            arrs["off_" + nm] = offs[i]
        arrs.update(coarse_raw_b0=rec.raw[0][:1], fine_raw_b0=rec.raw[1][:1],
                    coarse_order=rec.orders[0].to(torch.int16), fine_order=rec.orders[1].to(torch.int16),
                    range_fine_low=ranges[0]['low'], range_coarse_low=ranges[1]['low'])
    else:                # big case: outputs only, full-res map sub-sampled
        arrs.update(disp_full_sub4=disps[0][:, :, ::4, ::4], disp_precise=disps[1], disp_fine_up=disps[2],
                    disp_coarse_up=disps[3], cost_coarse=costs[2], samp_fine=samples[1],
                    disp_full_mean=disps[0].double().mean(), disp_full_absmean=disps[0].double().abs().mean())
    arrs.update(prev_disp_sub=info['prev_disp'][:, :, ::4, ::4],
                mem_out_disp_sample=info['cost_memory']['disp_sample'],
                mem_out_cost_volume=info['cost_memory']['cost_volume'])
    save(name, **arrs)


def reference_update_map():
    """The reference's OWN `update_map` (projects/TemporalStereo/TemporalStereo.py:326-461), compiled from its file's syntax tree
    (the LightningModule around it needs packages this image lacks); FunctionSoftsplat (CUDA-only) is bound to oracle.splat."""
    import ast
    import torch.nn.functional as F
    from architecture.modeling.layers import project_to_3d
    sys.path.insert(0, ROOT)
    from oracle import splat as osplat
    path = os.path.join(ref_import.REFERENCE_ROOT, "projects", "TemporalStereo", "TemporalStereo.py")
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TemporalStereo")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "update_map")
    ns = {"torch": torch, "F": F, "project_to_3d": project_to_3d, "EXPMAX": 50,
          "FunctionSoftsplat": lambda tenInput, tenFlow, tenMetric, strType: osplat.softsplat(tenInput, tenFlow, tenMetric, strType)}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["update_map"]


def planted_cases(M, only=None):
    """Full-size end-to-end fixtures: the REFERENCE aggregator with the committed contractive checkpoint
    (tests/golden/ckpt_planted.npz, trained by this repository's own TrainStep: tools/train_checkpoint.py) on planted-disparity
    scenes (tests/synth.stereo_sequence) at the BASELINE configurations' stated batches, sequences driven by the reference's own
    update_map.  Stored per frame: the full-resolution disparity sub-sampled, its EPE against the planted ground truth over ALL
    pixels (float64), the 1/4-resolution disparity, and a digest of the inputs so that the test side can tell whether it
    regenerated them bit for bit."""
    import types
    import parity_tools as PT
    update_map = reference_update_map()
    ckpt = PT.load_checkpoint()
    # (fixture name, configuration, seed offset, sub-sampling of the stored full-resolution map)
    plan = [("planted_c1_s0", 0, 0, 2), ("planted_c1_s1", 0, 1, 4), ("planted_c1_s2", 0, 2, 4),
            ("planted_c2_s0", 1, 0, 4), ("planted_c3_s0", 2, 0, 8), ("planted_c4_s0", 3, 0, 4),
            # round 4: a second reference-run seed for the temporal configurations, stored more sparsely
            ("planted_c2_s1", 1, 1, 8), ("planted_c3_s1", 2, 1, 12), ("planted_c4_s1", 3, 1, 6)]
    names = list(PT.CONFIGS)
    for name, ci, k, sub in plan:
        if only and name not in only:
            continue
        c = PT.CONFIGS[names[ci]]
        seed = synth.SEED0 + 500 + 10 * ci + k
        B, H, W, frames = c["B"], c["H"], c["W"], c["frames"]
        max_disp = 16 * c["num_sample"]
        dims = dict(SCENEFLOW); dims['coarse'] = dict(SCENEFLOW['coarse'], num_sample=c["num_sample"])
        net = build_reference_aggregator(dims)
        net.load_state_dict(ckpt, strict=True)
        net.eval()
        sc = synth.stereo_sequence(seed, B, H, W, frames=frames, max_disp=max_disp, fx=c["fx"], baseline=c["baseline"])
        me = types.SimpleNamespace(with_previous=True, use_past_cost=True, local_map_size=c["n_local"])
        eye = torch.eye(4).expand(B, 4, 4).contiguous()
        arrs = dict(config=names[ci], seed=seed, B=B, H=H, W=W, frames=frames, sub=sub, max_disp=max_disp, num_sample=c["num_sample"],
                    n_local=c["n_local"], fx=c["fx"], baseline=c["baseline"])
        info = {}
        for t in range(frames):
            lf, rf, il, ir = sc["frames"][t]
            arrs["input_checksum_%d" % t] = np.float64(synth.checksum([lf, rf, il, ir]))
            if t > 0:
                batch = {"baseline": torch.full((B, 1, 1, 1), float(c["baseline"])), ("color_aug", t, "l"): torch.zeros(B, 3, H, W),
                         ("K", 0): T(sc["K"]), ("inv_T", t - 1, "l"): eye, ("T", t, "l"): T(sc["T"][t])}
                with torch.no_grad():
                    _, info = update_map(me, batch, info, t)
            with torch.no_grad():
                disps, costs, samples, offs, ranges, info = net([T(x) for x in lf], [T(x) for x in rf], T(il), T(ir), info)
            gt = T(sc["gt"][t])
            arrs["epe_%d" % t] = np.float64(PT.epe(disps[0], gt, max_disp))
            arrs["epe_quarter_%d" % t] = np.float64(PT.epe(torch.nn.functional.interpolate(disps[1] * 4, size=(H, W), mode='bilinear', align_corners=True), gt, max_disp))
            arrs["disp_full_sub_%d" % t] = disps[0][:, :, ::sub, ::sub]
            arrs["disp_precise_sub_%d" % t] = disps[1][:, :, ::max(sub // 2, 1), ::max(sub // 2, 1)]
            arrs["disp_full_mean_%d" % t] = np.float64(disps[0].double().mean())
        arrs["mem_out_disp_sample"] = info['cost_memory']['disp_sample']
        if k == 0:            # (the later seeds stay below 1 MB: the test reads the candidates only)
            arrs["mem_out_cost_volume"] = info['cost_memory']['cost_volume']
        save(name, **arrs)
        print("   ", name, "EPE per frame", [float(arrs["epe_%d" % t]) for t in range(frames)])


def planted_gradient_case(M, full_size=False):
    """full_size: the same at BASELINE configs[1]'s size (544x960, D=192, B=1), stored as norms / projections only
    (planted_train_grads_c1, < 200 KB).
    Gradient fixture of the whole aggregator: the reference's aggregator in train() mode with the
    committed checkpoint on a small planted scene, the reference's OWN loss objects with the sceneflow.yaml weights behind the
    wrapper's full-resolution rescale (projects/TemporalStereo/TemporalStereo.py:305-309), loss.backward() by the framework's
    autograd -- run twice: in float32 (what the reference computes) and in float64 (the same modules and inputs cast to double:
    the exact gradient of the same function, the arbiter where fp32 backward passes through BatchNorm lose digits to
    cancellation).  Stored for both: the loss terms, the gradients of the six feature maps (every 8th channel), the gradients of
    a list of named weights, and for EVERY parameter the gradient's norm and its projection on a seeded random direction."""
    import torch.nn.functional as F
    import parity_tools as PT
    from architecture.modeling.losses import DispSmoothL1Loss, WarssersteinDistanceLoss
    B, H, W, ns = (1, 544, 960, 12) if full_size else (2, 128, 192, 4)
    max_disp = 16 * ns
    seed = synth.SEED0 + (610 if full_size else 600)
    fx = 1050.0 if full_size else 300.0
    dims = dict(SCENEFLOW); dims['coarse'] = dict(SCENEFLOW['coarse'], num_sample=ns)
    sc = synth.stereo_sequence(seed, B, H, W, frames=1, max_disp=max_disp, fx=fx, baseline=1.0)
    arrs = dict(seed=seed, B=B, H=H, W=W, num_sample=ns, max_disp=max_disp, fx=fx,
                input_checksum=np.float64(synth.checksum(list(sc["frames"][0]))))
    for tag, dt in (("", torch.float32), ("f64::", torch.float64)):
        torch.set_default_dtype(dt)                  # the reference creates constants (candidates, zero memory) in the default dtype
        net = build_reference_aggregator(dims)
        net.load_state_dict(PT.load_checkpoint(), strict=True)
        net = net.to(dt).train()
        lf, rf, il, ir = sc["frames"][0]
        lf = [T(x).to(dt).requires_grad_(True) for x in lf]
        rf = [T(x).to(dt).requires_grad_(True) for x in rf]
        gt = T(sc["gt"][0]).to(dt)
        disps, costs, samples, offs, ranges, info = net(lf, rf, T(il).to(dt), T(ir).to(dt), {})
        full = [F.interpolate(d * W / d.shape[-1], size=(H, W), mode='bilinear', align_corners=True) for d in disps]
        l1 = DispSmoothL1Loss(max_disp=max_disp, weights=[2.0, 1.0, 0.7, 0.5])(full, gt)
        wd = WarssersteinDistanceLoss(max_disp=max_disp, global_weight=2.0, weights=[1.0, 0.7, 0.5])(costs, offs, samples, gt)
        total = sum(l1.values()) + sum(wd.values())
        total.backward()
        arrs[tag + "total"] = total.detach()
        for k, v in list(l1.items()) + list(wd.items()):
            arrs[tag + "loss::" + k] = v.detach()
        for i in range(3):          # every eighth channel in full + norm and a seeded projection of the whole tensor
            for side, ts in (("left", lf), ("right", rf)):
                g = ts[i].grad
                if not full_size:
                    arrs[tag + "g_%s_%d" % (side, i)] = g[:, ::8]
                else:       # norms and a seeded projection only
                    pr = T(synth.normal(seed, "projf%s%d" % (side, i), tuple(g.shape))).double()
                    arrs[tag + "g_%s_%d_proj" % (side, i)] = np.float64((g.double() * pr).sum())
                arrs[tag + "g_%s_%d_norm" % (side, i)] = np.float64(g.double().norm())
        named = dict(net.named_parameters())
        picks = [k for k in named if k.endswith(("init3d.0.conv.0.weight", "init3d.0.conv.0.bias", "init3d.0.conv.1.weight", "past_conv.weight",
                                                 "pred_heads.cost_head.1.weight", "pred_heads.off_head.1.weight", "pred_heads.cost_head.0.weight",
                                                 "init3d.0.conv.1.norm.weight", "init3d.0.conv.1.norm.bias", "fuse.conv_5x5.weight"))]
        picks += [k for k in named if k.startswith("precise.refinement.") and k.endswith(("deconv4.weight", "deconv2.weight", "deconv2.bias", "conv4.0.weight",
                                                                                             "deconv4.norm.weight"))]
        picks += [k for k in named if "init3d.1." in k and k.endswith(".weight") and k.startswith("fine.")][:8]
        picks += [k for k in named if "convex_upsample" in k and k.endswith(".weight") and k.startswith("coarse.")][:4]
        picks = [k for k in dict.fromkeys(picks) if named[k].grad is not None and named[k].numel() <= 40000]
        if full_size:
            picks = [k for k in picks if named[k].numel() <= 600]          # a few small tensors element by element
        arrs["picked"] = np.array(picks)
        for k in picks:
            arrs[tag + "gw::" + k] = named[k].grad
        keys = sorted(k for k in named if named[k].grad is not None)
        arrs["all_keys"] = np.array(keys)
        arrs[tag + "all_norm"] = np.array([float(named[k].grad.double().norm()) for k in keys])
        arrs[tag + "all_proj"] = np.array([float((named[k].grad.double().flatten() * T(synth.normal(seed, "proj" + k, (named[k].numel(),))).double()).sum()) for k in keys])
        arrs["no_grad_keys"] = np.array(sorted(k for k in named if named[k].grad is None))
        print("    [%s] picked" % (tag or "f32"), len(picks), "of", len(keys), "parameters with gradients; total loss", float(total.detach()))
    torch.set_default_dtype(torch.float32)
    # how far the reference's fp32 backward is from its own exact (fp64) one: the floor of any fp32 comparison
    worst = 0.0
    for k in arrs["picked"]:
        a, b = arrs["gw::" + str(k)].double(), arrs["f64::gw::" + str(k)].double()
        worst = max(worst, float((a - b).norm() / b.norm().clamp_min(1e-30)))
    print("    reference fp32 vs fp64 backward, worst relative L2 over the picked weights: %.3g" % worst)
    save("planted_train_grads_c1" if full_size else "planted_train_grads", **arrs)


def temporal_update_cases(M):
    """K2c: the reference's OWN `update_map` (projects/TemporalStereo/TemporalStereo.py:326-461) executed here.

    The method lives inside a pytorch_lightning module whose import needs packages this image lacks, so its
    function definition is taken from the reference file's syntax tree and compiled as is (nothing is copied into
    this repository; the fixture holds inputs and outputs only).  It runs against the reference's real
    project_to_3d; the one foreign piece is FunctionSoftsplat, which exists only as cupy/CUDA kernels
    (softsplat.py:252,269-270) and is bound to oracle.splat.softsplat -- so these vectors pin the glue (intrinsics
    scaling, pose composition, disparity <-> depth, metric, plane selection, state bookkeeping), not the splat
    arithmetic, which stays pinned by the analytic tests of tests/test_oracle_splat.py."""
    import types
    update_map = reference_update_map()

    cases = [  # name, B, H, W, k, local maps in, local_map_size, use_past_cost, pre-composed pose, translation scale
        ("temporal_update_0", 2, 64, 96, 2, 2, 3, True, False, 1.0),
        ("temporal_update_1", 1, 72, 120, 2, 0, 3, True, False, 6.0),      # first temporal frame: no local map yet
        ("temporal_update_2", 2, 64, 96, 2, 3, 2, True, True, 1.0),        # cropped to local_map_size, T_past_to_now given
        ("temporal_update_3", 1, 80, 104, 3, 1, 0, True, False, 1.0),      # local maps off
        ("temporal_update_4", 2, 64, 96, 2, 2, 3, False, False, 1.0),      # past cost off
    ]
    for name, B, H, W, k, n_in, size, use_past, composed, tscale in cases:
        seed = synth.SEED0 + 300 + int(name[-1])
        h, w = H // 8, W // 8
        prev_disp = synth.uniform(seed, "pd", (B, 1, H, W), 2.0, 60.0)
        base = synth.uniform(seed, "mb", (B, 1, h, w), 1.0, 12.0)
        mem_s = (base + synth.uniform(seed, "ms", (B, k, h, w), -0.5, 0.5)).astype(np.float32)
        mem_c = synth.normal(seed, "mc", (B, k, h, w))
        lm = (base * synth.uniform(seed, "lm", (B, n_in, h, w), 0.9, 1.1)).astype(np.float32) if n_in else None
        K = synth.sceneflow_intrinsics(B, H, W)
        T_now = synth.small_motion(seed, B)
        T_now[:, :3, 3] *= tscale
        inv_T_past = np.linalg.inv(synth.small_motion(seed + 50, B)).astype(np.float32)
        baseline = synth.uniform(seed, "bl", (B, 1, 1, 1), 0.3, 1.2)
        me = types.SimpleNamespace(with_previous=True, use_past_cost=use_past, local_map_size=size)
        batch = {"baseline": T(baseline), ("color_aug", 0, "l"): torch.zeros(B, 3, H, W), ("K", 0): T(K),
                 ("inv_T", -1, "l"): T(inv_T_past), ("T", 0, "l"): T(T_now)}
        info = {"prev_disp": T(prev_disp), "cost_memory": {"disp_sample": T(mem_s), "cost_volume": T(mem_c)}}
        if lm is not None:
            info["local_map"] = T(lm)
        if composed:
            info["T_past_to_now"] = torch.bmm(T(T_now), T(inv_T_past))
        with torch.no_grad():
            outs, info = update_map(me, batch, info, 0)
        arrs = dict(prev_disp=prev_disp, mem_disp_sample=mem_s, mem_cost_volume=mem_c, K=K, T_now=T_now, inv_T_past=inv_T_past,
                    baseline=baseline, full_hw=np.array([H, W]), local_map_size=size, use_past_cost=int(use_past),
                    composed=int(composed), has_local_in=int(lm is not None), has_memory_out=int(info["cost_memory"] is not None),
                    has_local_out=int(size > 0))
        if lm is not None:
            arrs["local_map_in"] = lm
        if info["cost_memory"] is not None:
            arrs["out_disp_sample"] = info["cost_memory"]["disp_sample"]
            arrs["out_cost_volume"] = info["cost_memory"]["cost_volume"]
        if size > 0:
            arrs["out_local_map"] = info["local_map"]
        save(name, **arrs)


def loss_cases(M):
    """The reference's own loss objects (architecture/modeling/losses) on seeded outputs of the path's shape: three
    cost / offset / sample levels + four disparities at native resolution, rescaled to full size the way the model wrapper
    does (projects/TemporalStereo/TemporalStereo.py:305-309) before DispSmoothL1Loss sees them.  Values and autograd gradients."""
    import torch.nn.functional as F
    from architecture.modeling.losses import DispSmoothL1Loss, WarssersteinDistanceLoss
    cases = {"loss_dense": dict(hw=(64, 128), sparse=False, zeros=0.0, max_disp=48, levels=((16, 32, 5), (8, 16, 7), (4, 8, 14))),
             "loss_sparse": dict(hw=(64, 128), sparse=True, zeros=0.7, max_disp=48, levels=((16, 32, 5), (8, 16, 8), (4, 8, 14))),
             "loss_none_valid": dict(hw=(32, 64), sparse=False, zeros=1.0, max_disp=48, levels=((8, 16, 5), (4, 8, 7))),
             "loss_ragged": dict(hw=(50, 70), sparse=False, zeros=0.1, max_disp=64, levels=((13, 18, 5), (50, 70, 3)))}
    for ci, (name, c) in enumerate(cases.items()):
        seed = synth.SEED0 + 300 + ci
        B = 2
        H, W = c["hw"]
        gt = synth.uniform(seed, "gt", (B, 1, H, W), -4.0, 1.3 * c["max_disp"])
        gt = gt * (synth.uniform(seed, "keep", (B, 1, H, W)) >= c["zeros"])
        gt_t = torch.from_numpy(gt.astype(np.float32))
        arrs = dict(gt=gt, max_disp=c["max_disp"], sparse=int(c["sparse"]), n_levels=len(c["levels"]))
        costs, offs, samps, ests = [], [], [], []
        for li, (h, w, D) in enumerate(c["levels"]):
            costs.append(torch.from_numpy(synth.normal(seed, "cost%d" % li, (B, D, h, w), 2.0)).requires_grad_(True))
            offs.append(torch.from_numpy(synth.normal(seed, "off%d" % li, (B, D, h, w), 0.3)).requires_grad_(True))
            base = synth.uniform(seed, "base%d" % li, (B, 1, h, w), 0.0, c["max_disp"] * w / W)
            samps.append(torch.from_numpy((base + np.arange(D, dtype=np.float32).reshape(1, D, 1, 1) - D / 2).astype(np.float32)).requires_grad_(True))
            ests.append(torch.from_numpy(synth.uniform(seed, "est%d" % li, (B, 1, h, w), 0.0, c["max_disp"] * w / W)).requires_grad_(True))
        ests.append(torch.from_numpy(synth.uniform(seed, "estfull", (B, 1, H, W), 0.0, float(c["max_disp"]))).requires_grad_(True))
        wl = WarssersteinDistanceLoss(max_disp=c["max_disp"], sparse=c["sparse"])
        wd = wl(costs, offs, samps, gt_t)
        sl = DispSmoothL1Loss(max_disp=c["max_disp"], sparse=c["sparse"])
        full = [F.interpolate(d * W / d.shape[-1], size=(H, W), mode='bilinear', align_corners=True) for d in ests]
        sd = sl(full, gt_t)
        total = sum(wd.values()) + sum(sd.values())
        total.backward()
        for li in range(len(c["levels"])):
            arrs.update({"cost%d" % li: costs[li], "off%d" % li: offs[li], "sample%d" % li: samps[li],
                         "wars_loss%d" % li: wd["wars_loss_lvl%d" % li],
                         "g_cost%d" % li: costs[li].grad, "g_off%d" % li: offs[li].grad, "g_sample%d" % li: samps[li].grad})
        for li in range(len(ests)):
            arrs.update({"est%d" % li: ests[li], "l1_loss%d" % li: sd["l1_loss_lvl%d" % li],
                         "g_est%d" % li: ests[li].grad if ests[li].grad is not None else torch.zeros_like(ests[li]),
                         "full%d" % li: full[li]})
        arrs["n_est"] = len(ests)
        save(name, **arrs)


def backbone_memory_cases(M):
    """SURVEY 8(f)-4, last clause: the reference's OWN `_inverted_residual_forward`
    (architecture/modeling/backbone/TemporalStereo.py:183-218).  The module imports timm, which this image lacks, so the function's
    definition is taken from the reference file's syntax tree and compiled as is (nothing is copied; the fixture holds inputs and
    outputs only).  The timm block is replaced by identities and the values sit on a 1/8 grid, so that `out - input` is EXACTLY the
    tensor the block is fed -- these vectors pin the memory plumbing, which is all there is to pin."""
    import ast
    import types
    path = os.path.join(ref_import.REFERENCE_ROOT, "architecture", "modeling", "backbone", "TemporalStereo.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "_inverted_residual_forward")
    ns = {"torch": torch, "drop_path": lambda x, *a, **k: x}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    ref = ns["_inverted_residual_forward"]
    ident = lambda x: x
    block = types.SimpleNamespace(conv_pw=ident, bn1=ident, act1=ident, conv_dw=ident, bn2=ident, act2=ident, se=ident, conv_pwl=ident,
                                  bn3=ident, drop_path_rate=0.0, training=False)
    cases = [("feature_memory_0", 2, 24, 6, 10, 0.25, True), ("feature_memory_1", 2, 24, 6, 10, 0.25, False),
             ("feature_memory_2", 1, 10, 5, 7, 0.5, True), ("feature_memory_3", 3, 20, 4, 9, 0.125, True)]
    for name, B, C, H, W, pct, with_mem in cases:
        seed = synth.SEED0 + 400 + int(name[-1])
        grid = lambda tag, shape: (np.round(synth.uniform(seed, tag, shape, -8.0, 8.0) * 8.0) / 8.0).astype(np.float32)
        x = grid("x", (B, C, H, W))
        mc = int(C * pct)
        mem = grid("m", (B, mc, H, W)) if with_mem else None
        with torch.no_grad():
            out, new_mem = ref(block, T(x), T(mem) if mem is not None else None, pct)
        arrs = dict(input=x, memory_percent=np.float64(pct), has_memory=int(with_mem), out=out.numpy(), new_memory=new_mem.numpy())
        if mem is not None:
            arrs["memory"] = mem
        save(name, **arrs)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    M = ref_import.import_reference()
    if "--only-siblings" in sys.argv:
        sibling_cases(M)
        return
    if "--only-temporal" in sys.argv:
        temporal_update_cases(M)
        return
    if "--only-losses" in sys.argv:
        loss_cases(M)
        return
    if "--only-backbone-memory" in sys.argv:
        backbone_memory_cases(M)
        return
    if "--only-planted" in sys.argv:
        planted_cases(M, [a for a in sys.argv[1:] if a.startswith("planted_")] or None)
        if not any(a.startswith("planted_") for a in sys.argv[1:]):
            planted_gradient_case(M)
        return
    if "--only-planted-grads-c1" in sys.argv:
        planted_gradient_case(M, full_size=True)
        return
    if "--only-planted-grads" in sys.argv:
        planted_gradient_case(M)
        return
    functional_cases(M)
    sibling_cases(M)
    temporal_update_cases(M)
    loss_cases(M)
    backbone_memory_cases(M)
    planted_cases(M)
    planted_gradient_case(M)
    aggregator_case("agg_tiny_single", TINY, synth.SEED0 + 100, 2, 96, 160, temporal=False, store_inputs=True)
    aggregator_case("agg_tiny_temporal", TINY, synth.SEED0 + 101, 2, 96, 160, temporal=True, store_inputs=True)
    aggregator_case("agg_tiny_train", TINY, synth.SEED0 + 102, 2, 96, 160, temporal=False, store_inputs=True,
                    training=True)
    c1 = dict(SCENEFLOW); c1['coarse'] = dict(SCENEFLOW['coarse'], num_sample=3)      # D=48 -> 48/16
    aggregator_case("agg_config1_256x512", c1, synth.SEED0 + 1, 1, 256, 512, temporal=False, store_inputs=False)
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as fh:
        fh.write("Generated by tools/gen_golden.py from the reference at /root/reference (v1), torch %s CPU.\n"
                 "Inputs come from tests/synth.py seeds; weights from synth.state_values().\n" % torch.__version__)


if __name__ == "__main__":
    main()
