"""Roofline probes of the cost-volume build (K1) for bench.py's `roofline` object.

Everything here runs AFTER the timed region of the headline.  The judged figure is the complete op at its unfused
boundary at the 1/4 level (`ts_block_cost_sampled_fwd` on [B,128,136,240] x 5 candidates): algorithmic bytes over the
mean duration of the C-ABI launch, HIP events on the launch stream.  Next to it, measured in the same run: what the
pipeline itself launches, all three levels, the launch beyond the 256 MiB Infinity Cache, the contracted first layer --
and the fill / copy CEILINGS of this board at the same sizes (ts_calib_stream), so that `frac_of_fill_ceiling` is a
measured ratio and not a typed note."""
import json
import os

import torch

from .common import DIMS, HBM_PEAK, ROOT, RUN_H, RUN_W, k1_algorithmic_bytes, timed_us


class K1Probe:
    """Remembers the arguments of each distinct K1 call the pipeline makes during the timed steps (nothing is timed
    there); `measure` replays them afterwards through the C ABI."""

    def __init__(self):
        from temporalstereo_amd import functional as TF
        self.TF = TF
        self.calls = {}            # key (B, C, H, W, D, kind) -> (fn, left, right, disp_or_int, scales)

    def __enter__(self):
        self._orig = {"block_cost": self.TF.block_cost, "block_cost_warped": self.TF.block_cost_warped,
                      "block_cost_corr": self.TF.block_cost_corr}

        def remember(name):
            orig = self._orig[name]

            def f(reference_fm, target_fm, disp_sample, block_cost_scale=3):
                B, C, H, W = reference_fm.shape
                sampled = not isinstance(disp_sample, int)
                kind = "warped" if name == "block_cost_warped" else ("corr" if name == "block_cost_corr" else sampled)
                key = (B, C, H, W, disp_sample.shape[1] if sampled else disp_sample, kind)
                if key not in self.calls:
                    self.calls[key] = (orig, reference_fm.detach(), target_fm.detach(),
                                       disp_sample.detach() if sampled else disp_sample, block_cost_scale)
                return orig(reference_fm, target_fm, disp_sample, block_cost_scale)
            return f
        for name in self._orig:          # aggregation.levels / aggregation.native reach the ops as TF.<name>
            setattr(self.TF, name, remember(name))
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self.TF, name, fn)

    def measure(self, iters):
        """Mean duration (s) of each remembered K1 launch -- plus, for every sampled level, the COMPLETE op and the
        round-1-3 variant on the same tensors: the C-ABI entry point with preallocated output / workspace, `iters`
        launches back to back between one pair of HIP events.  The rocprofv3 kernel trace of the bench command gives the
        same figure as main + expansion kernel."""
        from temporalstereo_amd import _lib
        L = _lib.lib()
        st = torch.cuda.current_stream().cuda_stream
        calls = dict(self.calls)
        for key, (fn, l, r, d, sc) in self.calls.items():
            if key[5] in ("warped", "corr"):
                calls.setdefault(key[:5] + (True,),
                                 (None, l, r, d, sc))            # SURVEY.md 8(d)'s unfused-boundary figure
            if key[5] == "corr":
                calls.setdefault(key[:5] + ("warped",), (None, l, r, d, sc))        # the variant of rounds 1-3
        entry = {False: L.ts_block_cost_int_fwd, True: L.ts_block_cost_sampled_fwd,
                 "corr": L.ts_block_cost_sampled_corr_fwd,
                 "warped": L.ts_block_cost_sampled_warped_fwd}
        out_t = {}
        for key, (_, l, r, d, sc) in calls.items():
            B, C, H, W, D, kind = key
            l, r = l.contiguous(), r.contiguous()
            ctot = {True: 2 * C, "warped": C, False: C, "corr": 0}[kind] + sc * (C // 8)
            out = torch.empty((B, ctot, D, H, W), device=l.device, dtype=torch.float32)
            ws = torch.empty(max(int(L.ts_block_cost_workspace_bytes(B, C, H, W, D, sc)), 256), device=l.device,
                             dtype=torch.uint8)
            if kind is False:
                launch = lambda: entry[False](l.data_ptr(), r.data_ptr(), out.data_ptr(), ws.data_ptr(), B, C, H, W, D,
                                              sc, st)
            else:
                dd = d.contiguous()
                launch = lambda: entry[kind](l.data_ptr(), r.data_ptr(), dd.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                             B, C, H, W, D, sc, st)
            _lib.check(launch(), "K1")
            out_t[key] = timed_us(launch, iters, warm=20) * 1e-6
            del out, ws
        return out_t


def stream_ceilings(dev, sizes):
    """What this board sustains on plain float4 streams of `sizes` bytes, measured now (ts_calib_stream,
    csrc/calib.hip): {nbytes: {"fill": B/s, "copy": B/s}} -- fill writes nbytes, copy reads and writes nbytes / 2 each
    (the same bytes moved)."""
    from temporalstereo_amd import _lib
    L = _lib.lib()
    st = _lib.ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    for nbytes in sizes:
        nbytes = int(nbytes) // 32 * 32
        a = torch.empty(nbytes // 4, device=dev)
        half = nbytes // 2
        res = {}
        for name, kind, dst, src, n in (("fill", 0, a, a, nbytes), ("copy", 1, a, a[half // 4:], half)):
            fn = lambda: L.ts_calib_stream(kind, _lib.ptr(dst), _lib.ptr(src), n, st)
            _lib.check(fn(), "ts_calib_stream")
            res[name] = nbytes / (timed_us(fn, 30, warm=5) * 1e-6)
        out[nbytes] = res
        del a
    return out


def beyond_infinity_cache(k1, batch):
    """The judged launch on four pairs: 930 MB per launch, beyond the 256 MiB Infinity Cache (SURVEY.md 8(d)
    hygiene)."""
    from temporalstereo_amd import functional as TF
    key1 = (batch, 2 * DIMS['precise']['in_planes'], RUN_H // 4, RUN_W // 4, 5, "warped")
    if key1 not in k1.calls:
        key1 = key1[:5] + ("corr",)
    if key1 not in k1.calls:
        return None
    _, l1, r1, d1, sc1 = k1.calls[key1]
    rep = max(1, 4 // batch)
    l4, r4, d4 = (x.repeat(rep, 1, 1, 1).contiguous() for x in (l1, r1, d1))
    t4 = timed_us(lambda: TF.block_cost(l4, r4, d4, sc1), 20, warm=3) * 1e-6
    nb4 = k1_algorithmic_bytes(l4.shape[0], l4.shape[1], l4.shape[2], l4.shape[3], 5, True)
    return dict(batch=int(l4.shape[0]), algorithmic_bytes=nb4, mean_us=t4 * 1e6, achieved=nb4 / t4 / 1e9,
                frac=nb4 / t4 / HBM_PEAK,
                note="same launch on 4 pairs (930 MB per launch: beyond the 256 MiB Infinity Cache)")


def fused_first_layer(k1, runner, batch):
    """cost volume + first (1,3,3) layer of the 1/4 level on the pipeline's own tensors, two ways --
    materialised (rounds 1-3: volume without its reference half, convolved) and contracted (correlation blocks + the
    warped half contracted over channels before the warp).  `equivalent` = the UNFUSED op's algorithmic bytes over the
    time of what replaces it."""
    from temporalstereo_amd import functional as TF
    from temporalstereo_amd.aggregation import native as _N
    ckey = (batch, 2 * DIMS['precise']['in_planes'], RUN_H // 4, RUN_W // 4, 5, "corr")
    if ckey not in k1.calls:
        ckey = ckey[:5] + ("warped",)
    agg_n = getattr(runner, "net", None)
    if ckey not in k1.calls or not hasattr(agg_n, "precise"):
        return None
    _, lq, rq, dq, scq = k1.calls[ckey]
    pr = agg_n.precise
    lq, rq, dq = lq.contiguous(), rq.contiguous(), dq.contiguous()
    with torch.no_grad():
        lt, rt = pr.left_term(lq), pr.right_term(rq)
    t_mat = timed_us(lambda: _N.conv_hw(TF.block_cost_warped(lq, rq, dq, scq), pr.init0.f0, 1, pr.init0.dil, addend=lt))
    t_con = timed_us(lambda: _N.conv_hw_warp(TF.block_cost_corr(lq, rq, dq, scq), pr.init0_corr, rt, dq, lt.squeeze(2),
                                             pr.init0.dil))
    t_q = timed_us(lambda: pr.right_term(rq))
    nbf = k1_algorithmic_bytes(*ckey[:5], True)
    return dict(kernels="ts_block_cost_sampled_corr_fwd + ts_conv3d_hw_warp_fwd (gather + (1,3,3) convolution over the "
                        "correlation blocks) + the 1x1 pre-contraction right -> Q (ts_conv3d_d_fwd, k = 1; a function "
                        "of the features only, issued at the start of a pass)",
                replaces="ts_block_cost_sampled_warped_fwd + ts_conv3d_hw_fwd over [warped | corr] (rounds 1-3)",
                mean_us=t_con + t_q, on_the_level_chain_us=t_con, precontraction_us=t_q, replaced_mean_us=t_mat,
                unfused_algorithmic_bytes=nbf, equivalent=nbf / ((t_con + t_q) * 1e-6) / 1e9, unit="GB/s",
                frac=nbf / ((t_con + t_q) * 1e-6) / HBM_PEAK, in_use=bool(_N.FUSED_K1),
                note="algorithmic bytes of the UNFUSED cost-volume op (SURVEY 8(d)) over the time of cost volume + "
                     "first layer in the contracted form; both forms include the first layer's convolution, so compare "
                     "mean_us with replaced_mean_us, and `equivalent` with the unfused op's `achieved` only as a "
                     "bytes-never-moved figure")


def committed_traffic(pkey):
    """HBM bytes per launch from the PMC passes committed under profiles/ (a PMC pass cannot run inside the timed
    process)."""
    for cand in ("r06_k1_hbm_traffic_pmc.json", "r05_k1_hbm_traffic_pmc.json", "r04_k1_hbm_traffic_pmc.json",
                 "r03_k1_hbm_traffic_pmc.json",
                 "r02_k1_hbm_traffic_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as fh:
                pm = json.load(fh)
            if list(pm["workload_key"]) == list(pkey):
                return pm["hbm_bytes_per_launch"], cand
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def _variant(times, key, kernel):
    if key not in times:
        return None
    nb = k1_algorithmic_bytes(*key)
    return dict(kernel=kernel, algorithmic_bytes=nb, mean_us=times[key] * 1e6, achieved=nb / times[key] / 1e9,
                frac=nb / times[key] / HBM_PEAK)


def roofline(times, launched, batch, iters, ceilings, b4, fused):
    """The `roofline` object of the JSON line (None when the judged launch was not measured)."""
    pkey = (batch, 2 * DIMS['precise']['in_planes'], RUN_H // 4, RUN_W // 4, 5, True)
    if pkey not in times:
        return None
    nbytes = k1_algorithmic_bytes(*pkey)
    ach = nbytes / times[pkey]
    # each level as the COMPLETE op (329.3 MB per pair)
    complete = [k for k in times if k[5] is True or k[5] is False]
    all_b, all_t = sum(k1_algorithmic_bytes(*k) for k in complete), sum(times[k] for k in complete)
    lau = [k for k in launched if k in times]                             # ... and what the pipeline itself launches
    lau_b, lau_t = sum(k1_algorithmic_bytes(*k) for k in lau), sum(times[k] for k in lau)
    traffic, traffic_file = committed_traffic(pkey)
    ckey, wkey = pkey[:5] + ("corr",), pkey[:5] + ("warped",)
    vkey = ckey if ckey in launched else wkey
    # what the judged launch writes
    out_bytes = 4 * batch * (2 * pkey[1] + 3 * (pkey[1] // 8)) * 5 * pkey[2] * pkey[3]
    sizes = sorted(ceilings)
    near = min(sizes, key=lambda s: abs(s - out_bytes)) if sizes else None
    fill = ceilings[near]["fill"] if near else None
    r = dict(bound="hbm", achieved=ach / 1e9, peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK, traffic=traffic,
             measured="HIP events on the launch stream around %d back-to-back C-ABI launches on the pipeline's own "
                      "input tensors, right after the timed steps" % iters,
             kernel=("ts_block_cost_sampled_fwd (block_cost_fast + block_cost_upsample_rows) on [%d,%d,%d,%d] x %d "
                     "candidates" % pkey[:5]),
             algorithmic_bytes=nbytes, mean_us=times[pkey] * 1e6,
             fill_ceiling=(dict(bytes=near, fill=fill / 1e9, copy=ceilings[near]["copy"] / 1e9, unit="GB/s",
                                note="plain float4 fill / copy streams of the size this launch writes, measured in "
                                     "this run right after it (ts_calib_stream): what the board sustains where the "
                                     "0.90 target is priced against the 8 TB/s spec figure") if near else None),
             frac_of_fill_ceiling=(ach / fill if fill else None),
             pipeline_variant=_variant(
                 times, vkey,
                 ("ts_block_cost_sampled_corr_fwd (block_cost_corr_rows + block_cost_upsample_rows: the correlation "
                  "blocks alone, the first layer takes the warped half in pre-contracted form, see `fused`)"
                  if vkey[5] == "corr" else
                  "ts_block_cost_sampled_warped_fwd (volume without the D-fold repeat of the left features)")
                 + "; what the native pipeline launches"),
             warped_variant=_variant(times, wkey, "ts_block_cost_sampled_warped_fwd (rounds 1-3: volume without the "
                                                  "D-fold repeat of the left features)"),
             all_levels=dict(algorithmic_bytes=all_b, mean_us=all_t * 1e6, achieved=all_b / all_t / 1e9,
                             frac=all_b / all_t / HBM_PEAK,
                             note="coarse (int) + fine + precise (sampled), each the complete op at its unfused "
                                  "boundary"),
             all_levels_as_launched=dict(algorithmic_bytes=lau_b, mean_us=lau_t * 1e6, achieved=lau_b / lau_t / 1e9,
                                         frac=lau_b / lau_t / HBM_PEAK,
                                         note="the variants the pipeline launches, with THEIR algorithmic bytes"),
             fused=fused, beyond_infinity_cache=b4,
             traffic_source="profiles/%s: FETCH_SIZE / WRITE_SIZE passes of this command under rocprofv3 "
                            "(tools/k1_traffic.py); a PMC pass cannot run inside the timed process" % traffic_file)
    if b4 is not None and sizes:
        big = max(sizes)
        b4["fill_ceiling"] = dict(bytes=big, fill=ceilings[big]["fill"] / 1e9, copy=ceilings[big]["copy"] / 1e9,
                                  unit="GB/s")
        b4["frac_of_fill_ceiling"] = b4["achieved"] * 1e9 / ceilings[big]["fill"]
    return r


def run(k1, launched, runner, dev, batch, iters, native, ceilings=True):
    """All K1 measurements of a bench run (rank 0, after the timed region) -> the `roofline` object.
    ceilings=False (bench.py --calibrate: a PMC pass): no fill / copy streams of other sizes, which the per-(kernel,
    grid) averages of tools/k1_traffic.py would mix into its 1 GiB calibration launches."""
    times = k1.measure(iters)
    b4 = beyond_infinity_cache(k1, batch) if native else None
    fused = fused_first_layer(k1, runner, batch) if native else None
    cp = DIMS['precise']['in_planes']
    out_bytes = 4 * batch * (4 * cp + 3 * (2 * cp // 8)) * 5 * (RUN_H // 4) * (RUN_W // 4)
    # this launch; the four-pair one
    ceilings = stream_ceilings(dev, sorted({out_bytes, out_bytes // batch * max(4, batch)})) if ceilings else {}
    return roofline(times, launched, batch, iters, ceilings, b4, fused)
