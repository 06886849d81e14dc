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
