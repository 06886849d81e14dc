"""Oracle (test infrastructure): rigid re-projection, K2a of SURVEY.md section 8(a)."""
import torch


def project_to_3d(depth, K, inv_K=None, T_target_to_source=None, eps=1e-7):
    """Back-project depth planes, move them by T, re-project.

    Follows architecture/modeling/layers/inverse_warp.py:92-178 (same dict keys).
    depth [B,C,H,W]; K,inv_K [B,3|4,3|4]; T [B,4,4].
    """
    B, C, H, W = depth.shape
    dt = depth.dtype
    xs = torch.arange(0, W, dtype=dt).view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(0, H, dtype=dt).view(1, 1, H, 1).expand(B, 1, H, W)
    pix = torch.cat((xs, ys), dim=1)                                        # mesh_grid :80-90
    homo = torch.cat((pix, torch.ones(B, 1, H, W, dtype=dt)), dim=1)
    homo = homo.reshape(B, 3, -1).repeat(1, 1, C)                           # :126
    z = depth.reshape(B, 1, -1)                                             # :128
    if inv_K is None:
        inv_K = torch.inverse(K[:, :3, :3])
    pts = torch.matmul(inv_K[:, :3, :3], homo) * z                          # :132
    hpts = torch.cat((pts, torch.ones(B, 1, C * H * W, dtype=dt)), dim=1)
    out = {'homo_points_3d': hpts}
    if T_target_to_source is None:
        return out
    if K.shape[-1] == 3:                                                    # :138-143
        K4 = torch.eye(4, dtype=dt).unsqueeze(0).repeat(B, 1, 1)
        K4[:, :3, :3] = K[:, :3, :3]
    else:
        K4 = K
    P = torch.matmul(K4, T_target_to_source)[:, :3, :]
    cam = torch.matmul(P, hpts)                                             # :148
    out['triangular_depth'] = cam[:, -1, :].reshape(B, C, H, W).contiguous()
    uv = cam[:, :2, :] / (cam[:, 2:3, :] + eps)                             # :154
    uv = uv.reshape(B, 2, C, H, W).permute(0, 2, 1, 3, 4).contiguous()
    ok = (uv[:, :, 0:1] >= 0) & (uv[:, :, 0:1] <= W - 1) & (uv[:, :, 1:2] >= 0) & (uv[:, :, 1:2] <= H - 1)
    out['flow_mask'] = ok.reshape(B, C, H, W).contiguous()
    uv = uv.reshape(B, C * 2, H, W).contiguous()
    out['src_pixel_coord'] = uv
    out['optical_flow'] = uv - pix.repeat(1, C, 1, 1)                       # :170
    return out
