"""Dense, per-pixel, autograd-differentiable restatement of the splatting equations (torch, float64).

Independent of both oracle/gp_oracle.c and the HIP kernels: no tiles lists, no sort-by-key, no
hand-derived backward -- every pixel looks at every Gaussian, ordering is a plain argsort by depth,
gradients come from torch.autograd.  It is used to validate the ORACLE's hand-written backward (and
forward) at small sizes.  Same declared semantics as the oracle (near plane 0.2, 16x16 tile rects
decide which pixels a splat may touch, +0.3 dilation, alpha=min(0.99,o*G) with straight-through
gradient, skip alpha<1/255, stop when T would drop below 1e-4).
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_basis(deg, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)  # [N, (deg+1)^2]


def quat_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def dense_render(means3D, means2D_ndc, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                 viewmatrix, projmatrix, campos, bg, H, W, tanfovx, tanfovy, sh_degree, scale_modifier=1.0):
    """All tensors float64.  viewmatrix/projmatrix are the reference's row-vector (transposed) matrices.
    Returns image[3,H,W], depth[H,W], radii[N]."""
    N = means3D.shape[0]
    V = viewmatrix.t()   # standard: p_view = V @ [p;1]
    P = projmatrix.t()
    hom = torch.cat([means3D, torch.ones(N, 1, dtype=means3D.dtype)], dim=1)
    pv = hom @ V.t()
    ph = hom @ P.t()
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None] + means2D_ndc[:, :2]
    tz = pv[:, 2]
    in_front = tz > 0.2
    if cov3D_precomp is None:
        Rm = quat_R(rotations)
        L = Rm * (scale_modifier * scales)[:, None, :]
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            dim=1).reshape(-1, 3, 3)
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tzs = torch.where(in_front, tz, torch.ones_like(tz))
    txtz, tytz = pv[:, 0] / tzs, pv[:, 1] / tzs
    inx = (txtz >= -limx) & (txtz <= limx)
    iny = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(inx, pv[:, 0], (txtz.clamp(-limx, limx) * tzs).detach())
    ty = torch.where(iny, pv[:, 1], (tytz.clamp(-limy, limy) * tzs).detach())
    zero = torch.zeros_like(tzs)
    J = torch.stack([fx / tzs, zero, -fx * tx / (tzs * tzs), zero, fy / tzs, -fy * ty / (tzs * tzs)], dim=1).reshape(-1, 2, 3)
    Wm = V[:3, :3]
    T = J @ Wm
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    dets = torch.where(ok, det, torch.ones_like(det))
    conA, conB, conC = c / dets, -b / dets, a / dets
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    rad = torch.ceil(3 * torch.sqrt(lam)).detach()
    pix = ((ndc[:, 0] + 1) * W - 1) * 0.5
    piy = ((ndc[:, 1] + 1) * H - 1) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    minx = torch.trunc((pix.detach() - rad) / 16).clamp(0, gx)
    miny = torch.trunc((piy.detach() - rad) / 16).clamp(0, gy)
    maxx = torch.trunc((pix.detach() + rad + 15) / 16).clamp(0, gx)
    maxy = torch.trunc((piy.detach() + rad + 15) / 16).clamp(0, gy)
    ok = ok & ((maxx - minx) * (maxy - miny) > 0)
    radii = torch.where(ok, rad, torch.zeros_like(rad)).to(torch.int32)
    if colors_precomp is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        Bm = sh_basis(sh_degree, d)
        col = torch.einsum("nk,nkc->nc", Bm, shs[:, :Bm.shape[1], :]) + 0.5
        col = torch.clamp_min(col, 0.0)
    else:
        col = colors_precomp
    # ordering: depth ascending, ties by index (stable)
    order = torch.argsort(tz.detach(), stable=True)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=means3D.dtype), torch.arange(W, dtype=means3D.dtype), indexing="ij")
    pxf, pyf = xs.reshape(-1), ys.reshape(-1)
    ptx, pty = torch.floor(pxf / 16), torch.floor(pyf / 16)
    o = order
    dx = pix[o][None, :] - pxf[:, None]
    dy = piy[o][None, :] - pyf[:, None]
    power = -0.5 * (conA[o][None] * dx * dx + conC[o][None] * dy * dy) - conB[o][None] * dx * dy
    member = ok[o][None] & (ptx[:, None] >= minx[o][None]) & (ptx[:, None] < maxx[o][None]) & \
        (pty[:, None] >= miny[o][None]) & (pty[:, None] < maxy[o][None])
    G = torch.exp(torch.clamp(power, max=0.0))
    alpha_raw = opacities.reshape(-1)[o][None] * G
    alpha = alpha_raw + (torch.clamp(alpha_raw, max=0.99) - alpha_raw).detach()
    valid = member & (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    alpha_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    T_excl = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1, dtype=alpha.dtype), 1 - alpha_eff[:, :-1]], dim=1), dim=1)
    test_T = T_excl * (1 - alpha_eff)
    stop = (valid & (test_T.detach() < 1e-4)).to(torch.int32).cumsum(dim=1) > 0
    alpha_fin = torch.where(stop, torch.zeros_like(alpha_eff), alpha_eff)
    T_ex = torch.cumprod(torch.cat([torch.ones(alpha.shape[0], 1, dtype=alpha.dtype), 1 - alpha_fin[:, :-1]], dim=1), dim=1)
    w = alpha_fin * T_ex
    T_final = torch.prod(1 - alpha_fin, dim=1)
    img = w @ col[o] + T_final[:, None] * bg[None]
    depth = w @ tz[o]
    wmax, arg = w.max(dim=1)
    tidx = torch.where(wmax > 0, o[arg], torch.full_like(arg, -1))
    return img.t().reshape(3, H, W), depth.reshape(H, W), radii, tidx.reshape(H, W).to(torch.int32)
