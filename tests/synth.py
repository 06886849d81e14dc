"""Deterministic synthetic inputs shared by tools/gen_golden.py (reference side, build container)
and the tests / bench (product side).  Pure numpy `RandomState` (frozen legacy stream) so the
same seed gives the same bytes on every machine; nothing is read from the reference checkout.

Shapes follow SURVEY.md section 8(d): unit-variance, spatially smooth features at 1/4, 1/8, 1/16 of
the run resolution (low-passed N(0,1), see `smooth`), images likewise, weights by the reference initialiser's distribution
(normal(0, sqrt(2/(k*Cout))), coarse.py:52-67) and BatchNorm buffers randomised so that
folding is exercised.
"""
import zlib

import numpy as np

SEED0 = 20260928


def _rs(seed, tag=""):
    return np.random.RandomState((int(seed) + zlib.crc32(tag.encode())) % (2 ** 32))


def normal(seed, tag, shape, scale=1.0):
    return (_rs(seed, tag).standard_normal(size=shape) * scale).astype(np.float32)


def uniform(seed, tag, shape, lo=0.0, hi=1.0):
    return _rs(seed, tag).uniform(lo, hi, size=shape).astype(np.float32)


def smooth(a, passes=2):
    """Separable binomial [1,4,6,4,1]/16 low-pass along the last two axes (edge-replicated), applied
    `passes` times and rescaled to unit variance: CNN feature maps are spatially smooth; white noise
    would make every sub-pixel resampling maximally ill-conditioned (d feature / d disparity ~ 1.4)."""
    k = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16.0
    x = a.astype(np.float64)
    for _ in range(passes):
        for ax in (-2, -1):
            pad = [(0, 0)] * x.ndim
            pad[ax] = (2, 2)
            xp = np.pad(x, pad, mode="edge")
            n = x.shape[ax]
            x = sum(k[i] * np.take(xp, np.arange(i, i + n), axis=ax) for i in range(5))
    x = x / x.std()
    return x.astype(np.float32)


def feature_pyramid(seed, B, H, W, chans=(64, 128, 256), correlated=True, max_shift=6.0):
    """[f4, f8, f16] left and right pyramids for a run resolution H x W (both multiples of 16).

    With `correlated`, the right map is the left map shifted by a smooth positive disparity plus
    noise, so that the matching problem is not pure noise (still fully synthetic).
    """
    lefts, rights = [], []
    for lvl, (c, s) in enumerate(zip(chans, (4, 8, 16))):
        h, w = H // s, W // s
        L = smooth(normal(seed, "L%d" % lvl, (B, c, h, w)))
        if correlated:
            ys = np.linspace(0, 1, h, dtype=np.float32).reshape(1, 1, h, 1)
            xs = np.arange(w, dtype=np.float32).reshape(1, 1, 1, w)
            d = (max_shift / (2 ** lvl)) * (0.35 + 0.65 * ys) + 0 * xs      # [1,1,h,w]
            src = xs + d                                                     # R[x] = L[x + d]
            x0 = np.floor(src).astype(np.int64)
            fr = (src - x0).astype(np.float32)
            x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
            x0b = np.broadcast_to(x0c, L.shape)
            x1b = np.broadcast_to(x1c, L.shape)
            R = (1 - fr) * np.take_along_axis(L, x0b, axis=3) + fr * np.take_along_axis(L, x1b, axis=3)
            R = (R + 0.1 * smooth(normal(seed, "Rn%d" % lvl, L.shape))).astype(np.float32)
        else:
            R = smooth(normal(seed, "R%d" % lvl, (B, c, h, w)))
        lefts.append(L)
        rights.append(R)
    return lefts, rights


def images(seed, B, H, W):
    return smooth(normal(seed, "imgL", (B, 3, H, W)), 3), smooth(normal(seed, "imgR", (B, 3, H, W)), 3)


def state_values(shapes, seed):
    """Deterministic parameter/buffer values for a module.

    shapes: {state_dict key: tuple shape}.  Returns {key: np.ndarray}.  Distribution by key suffix.
    """
    out = {}
    for key in sorted(shapes):
        shp = tuple(int(s) for s in shapes[key])
        leaf = key.rsplit(".", 1)[-1]
        owner = key.rsplit(".", 1)[0] if "." in key else ""
        is_bn = owner.endswith(".norm") or owner.endswith("mask.1")
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shp, dtype=np.int64)
        elif leaf == "running_mean":
            out[key] = normal(seed, key, shp, 0.1)
        elif leaf == "running_var":
            out[key] = uniform(seed, key, shp, 0.5, 1.5)
        elif is_bn and leaf == "weight":
            out[key] = uniform(seed, key, shp, 0.5, 1.5)
        elif is_bn and leaf == "bias":
            out[key] = normal(seed, key, shp, 0.1)
        elif leaf == "weight" and len(shp) >= 3:
            fan = int(np.prod(shp[2:])) * shp[0]
            # ConvTranspose weights are [Cin, Cout, ...]; scale choice only needs to be deterministic
            out[key] = normal(seed, key, shp, float(np.sqrt(2.0 / fan)))
        elif leaf == "bias":
            out[key] = normal(seed, key, shp, 0.05)
        elif leaf == "phi":
            out[key] = np.zeros(shp, dtype=np.float32)
        else:
            out[key] = normal(seed, key, shp, 0.1)
    return out


def sceneflow_intrinsics(B, H, W):
    """fx=fy=1050, cx=497.5, cy=269.5 at 540x960 (datasets/scene_flow/base.py:15-18, normalised),
    scaled to the run size; 4x4."""
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = 1050.0 * W / 960.0
    K[1, 1] = 1050.0 * H / 540.0
    K[0, 2] = 497.5 * W / 960.0
    K[1, 2] = 269.5 * H / 540.0
    return np.broadcast_to(K, (B, 4, 4)).copy()


def small_motion(seed, B, max_deg=1.0, max_t=0.1):
    """T_past_to_now: rotation <= max_deg about a random axis + translation <= max_t (metres)."""
    rs = _rs(seed, "pose")
    T = np.zeros((B, 4, 4), dtype=np.float64)
    for b in range(B):
        axis = rs.standard_normal(3)
        axis /= np.linalg.norm(axis)
        ang = np.deg2rad(rs.uniform(-max_deg, max_deg))
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * (Kx @ Kx)
        T[b, :3, :3] = R
        T[b, :3, 3] = rs.uniform(-max_t, max_t, size=3)
        T[b, 3, 3] = 1.0
    return T.astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------
# Planted-disparity stereo scenes (round 3).  Inputs that ARE stereo: the left maps are the right maps resampled along
# x by a known left-view disparity field, at every pyramid level and in the images, and over a sequence the field moves
# rigidly with the camera poses.  A network trained on these (tools/train_checkpoint.py) is contractive, which is what
# makes |dEPE| < 1e-3 px assertable per seed and per frame (tests/test_fullsize_gpu.py).  Only +, *, floor and table
# look-ups on top of the RandomState streams: no transcendental whose last bit depends on the host's SIMD level.
_BINOMIAL = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16.0


def _smooth_gain(passes):
    """Standard deviation of white N(0,1) noise after `passes` separable binomial passes (interior pixels)."""
    k = np.array([1.0])
    for _ in range(passes):
        k = np.convolve(k, _BINOMIAL)
    return float((k ** 2).sum())            # 1-D factor squared = product of the two axes' factors (each sqrt(sum k^2))


def smooth_fixed(a, passes=2):
    """`smooth` with an analytic unit-variance gain instead of a data-dependent one (float32 throughout)."""
    k = _BINOMIAL.astype(np.float32)
    x = a.astype(np.float32)
    for _ in range(passes):
        for ax in (-2, -1):
            pad = [(0, 0)] * x.ndim
            pad[ax] = (2, 2)
            xp = np.pad(x, pad, mode="edge")
            n = x.shape[ax]
            idx = [slice(None)] * x.ndim
            acc = None
            for i in range(5):
                idx[ax] = slice(i, i + n)
                term = k[i] * xp[tuple(idx)]
                acc = term if acc is None else acc + term
            x = acc
    return (x * np.float32(1.0 / _smooth_gain(passes))).astype(np.float32)


def _control_field(rs, gh, gw, H, W, amp):
    """Piecewise-bilinear field [H,W] through a gh x gw grid of control values ~ U(-amp, amp) (float64)."""
    ctrl = rs.uniform(-amp, amp, size=(gh, gw))
    y = np.arange(H, dtype=np.float64) * ((gh - 1) / max(H - 1, 1))
    x = np.arange(W, dtype=np.float64) * ((gw - 1) / max(W - 1, 1))
    y0 = np.minimum(np.floor(y).astype(np.int64), gh - 2); fy = (y - y0)[:, None]
    x0 = np.minimum(np.floor(x).astype(np.int64), gw - 2); fx = (x - x0)[None, :]
    c00, c01 = ctrl[y0][:, x0], ctrl[y0][:, x0 + 1]
    c10, c11 = ctrl[y0 + 1][:, x0], ctrl[y0 + 1][:, x0 + 1]
    return (1 - fy) * ((1 - fx) * c00 + fx * c01) + fy * ((1 - fx) * c10 + fx * c11)


def pinhole(B, H, W, fx):
    """4x4 intrinsics with the principal point at the image centre (tests/parity_tools.intrinsics)."""
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = fx
    K[0, 2], K[1, 2] = W / 2 - 0.5, H / 2 - 0.5
    return np.broadcast_to(K, (B, 4, 4)).copy()


def _plane_disparity(n, h, K, fb, H, W):
    """Disparity of the plane n.X = h seen through K: d = fb / Z with 1/Z = n.K^-1[x,y,1] / h -- affine in (x, y)."""
    fx, fy, cx, cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    a, b = fb * n[0] / (h * fx), fb * n[1] / (h * fy)
    c = fb * n[2] / h - a * cx - b * cy
    return a * np.arange(W, dtype=np.float64)[None, :] + b * np.arange(H, dtype=np.float64)[:, None] + c


def _warp_rows(src, shift):
    """out[..., y, x] = src[..., y, x - shift[y, x]] (linear interpolation, edge-replicated), float32."""
    w = src.shape[-1]
    pos = np.arange(w, dtype=np.float32)[None, :] - shift.astype(np.float32)
    x0 = np.floor(pos)
    fr = (pos - x0).astype(np.float32)
    x0 = x0.astype(np.int64)
    i0 = np.broadcast_to(np.clip(x0, 0, w - 1), src.shape)
    i1 = np.broadcast_to(np.clip(x0 + 1, 0, w - 1), src.shape)
    return ((1 - fr) * np.take_along_axis(src, i0, axis=-1) + fr * np.take_along_axis(src, i1, axis=-1)).astype(np.float32)


def stereo_sequence(seed, B, H, W, frames=1, max_disp=192, fx=None, baseline=1.0, chans=(64, 128, 256), noise=0.05, bumps=True,
                    max_deg=1.0, max_t=0.1):
    """A rigid planted-disparity scene seen over `frames` frames, oldest first.

    Returns dict(frames=[(left_feats, right_feats, left_image, right_image)], gt=[[B,1,H,W] left-view disparity per frame],
                 K=[B,4,4], T=[frames x [B,4,4]] with T[t] the pose change frame t-1 -> frame t (T[0] = identity)).
    Per item: a slanted plane (disparity 0.06 ... 0.75 of max_disp over the image) that every frame sees through its own pose,
    plus a frame-independent piecewise-bilinear relief of a few pixels; features at 1/4, 1/8, 1/16 and the images are smooth
    noise with left(x) = right(x - d(x)) + `noise` x independent smooth noise on both views.
    """
    fx = float(fx if fx is not None else 1050.0 * W / 960.0)
    fb = fx * float(baseline)
    K = pinhole(B, H, W, fx)
    T = [np.broadcast_to(np.eye(4, dtype=np.float32), (B, 4, 4)).copy()] + [small_motion(seed + 31 * t, B, max_deg, max_t) for t in range(1, frames)]
    gts = [np.zeros((B, 1, H, W), dtype=np.float32) for _ in range(frames)]
    for b in range(B):
        rs = _rs(seed, "scene%d" % b)
        centre = rs.uniform(0.25, 0.45) * max_disp
        dx, dy = rs.uniform(-0.12, 0.12) * max_disp, rs.uniform(0.0, 0.22) * max_disp       # change across the whole width / height
        a, bb = dx / (W - 1), dy / (H - 1)
        c = centre - a * (W - 1) / 2 - bb * (H - 1) / 2
        k = K[b].astype(np.float64)
        n = np.array([a * k[0, 0], bb * k[1, 1], c + a * k[0, 2] + bb * k[1, 2]]) / fb
        h = 1.0
        relief = _control_field(rs, 5, 8, H, W, 0.02 * max_disp) if bumps else 0.0
        for t in range(frames):
            if t > 0:
                Tt = T[t][b].astype(np.float64)
                n = Tt[:3, :3] @ n
                h = h + float(n @ Tt[:3, 3])
            d = _plane_disparity(n, h, k, fb, H, W) + relief
            gts[t][b, 0] = np.clip(d, 0.02 * max_disp, 0.9 * max_disp).astype(np.float32)
    out_frames = []
    for t in range(frames):
        lefts, rights = [], []
        for lvl, (cch, s) in enumerate(zip(chans, (4, 8, 16))):
            hh, ww = H // s, W // s
            yi = np.round(np.arange(hh, dtype=np.float64) * ((H - 1) / max(hh - 1, 1))).astype(np.int64)
            xi = np.round(np.arange(ww, dtype=np.float64) * ((W - 1) / max(ww - 1, 1))).astype(np.int64)
            L = np.empty((B, cch, hh, ww), dtype=np.float32)
            R = np.empty((B, cch, hh, ww), dtype=np.float32)
            for b in range(B):
                tag = "f%d_l%d_b%d" % (t, lvl, b)
                base = smooth_fixed(normal(seed, "R" + tag, (cch, hh, ww)))
                shift = gts[t][b, 0][yi][:, xi] * np.float32(ww / W)
                L[b] = _warp_rows(base, shift) + np.float32(noise) * smooth_fixed(normal(seed, "nl" + tag, (cch, hh, ww)))
                R[b] = base + np.float32(noise) * smooth_fixed(normal(seed, "nr" + tag, (cch, hh, ww)))
            lefts.append(L)
            rights.append(R)
        iL = np.empty((B, 3, H, W), dtype=np.float32)
        iR = np.empty((B, 3, H, W), dtype=np.float32)
        for b in range(B):
            tag = "f%d_b%d" % (t, b)
            base = smooth_fixed(normal(seed, "iR" + tag, (3, H, W)), 3)
            iL[b] = _warp_rows(base, gts[t][b, 0]) + np.float32(noise) * smooth_fixed(normal(seed, "inl" + tag, (3, H, W)), 3)
            iR[b] = base + np.float32(noise) * smooth_fixed(normal(seed, "inr" + tag, (3, H, W)), 3)
        out_frames.append((lefts, rights, iL, iR))
    return dict(frames=out_frames, gt=gts, K=K, T=T)


def checksum(arrays):
    """Order-sensitive float64 digest of a nest of arrays: the fixtures carry it so that a test can tell "the inputs were not
    regenerated bit for bit on this host" from "the kernels differ"."""
    acc, n = 0.0, 0
    stack = [arrays]
    while stack:
        a = stack.pop()
        if isinstance(a, (list, tuple)):
            stack.extend(reversed(a))
            continue
        v = np.asarray(a, dtype=np.float64).ravel()
        wts = (np.arange(v.size, dtype=np.float64) % 251.0) + 1.0
        acc += float(np.add.reduce(v * wts)) * (1.0 + 0.001 * n)
        n += 1
    return acc
