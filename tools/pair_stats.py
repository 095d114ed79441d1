#!/usr/bin/env python
"""Diagnostic: how many (pixel, splat) pairs of the bench workload are candidates at each culling granularity
(whole 16x16 tile, 16x8 half, 8x8 quadrant, 16x2 row pair, 8x2, exact alpha >= 1/255), counted up to each pixel's
n_contrib (early termination).  Sampled tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from types import SimpleNamespace
from gaussianprediction_amd.rasterizer import raster_forward_debug
from gaussianprediction_amd.renderer import _settings
args = SimpleNamespace(gaussians=1_000_000, width=1352, height=1014, keypoints=250, nearest_num=6, time_freq=8, iteration=50000,
                       scale_lo=0.003, scale_hi=0.012)
dev = torch.device("cuda", 0)
pc, cams, gts, margs = bench.build_workload(args, dev)
W, H = 1352, 1014
gx = (W + 15) // 16
rng = np.random.default_rng(0)
with torch.no_grad():
    cam = cams[3]
    t = torch.from_numpy(cam.time).float().to(dev)
    xyz, q, s, o = pc(t, 50000)
    dbg = raster_forward_debug(_settings(cam, pc, torch.zeros(3, device=dev), 1.0), xyz, o, shs=pc.get_features, scales=s, rotations=q)
    rec, pl, ranges, nc = dbg["rec"], dbg["point_list"].long(), dbg["ranges"].cpu().numpy(), dbg["n_contrib"]
    tot = dict(tile=0, half=0, quad=0, row2=0, q8x2=0, q4x4=0, exact=0, lens=0, halfvis=0, quadvis=0)
    tiles = rng.choice(len(ranges), size=400, replace=False)
    for tl in tiles:
        a, b = ranges[tl]
        if b <= a: continue
        ids = pl[a:b]
        r = rec[ids]                                      # [L,12]
        # record (round 4): (x, y, A', B') (C', log2 opacity, depth, id) (r, g, b, opacity); A' B' C' = log2(e) x the quadratic form
        LN2 = 0.6931471805599453
        x, y, A, B, Cc, op = r[:, 0], r[:, 1], r[:, 2] * LN2, r[:, 3] * LN2, r[:, 4] * LN2, r[:, 11]
        tx, ty = tl % gx, tl // gx
        px = (torch.arange(16, device=dev) + 16 * tx).float()
        py = (torch.arange(16, device=dev) + 16 * ty).float()
        dx = px[None, None, :] - x[:, None, None]          # [L,1,16]   (sign conventions do not matter for the count)
        dy = py[None, :, None] - y[:, None, None]          # [L,16,1]
        power = dx * (A[:, None, None] * dx + B[:, None, None] * dy) + (Cc[:, None, None] * dy) * dy
        alpha = torch.clamp(op[:, None, None] * torch.exp(power), max=0.99)
        inimg = (px[None, None, :] < W) & (py[None, :, None] < H)
        ncl = nc[16 * ty:16 * ty + 16, 16 * tx:16 * tx + 16]
        if ncl.shape != (16, 16):
            pad = torch.zeros(16, 16, dtype=ncl.dtype, device=dev); pad[:ncl.shape[0], :ncl.shape[1]] = ncl; ncl = pad
        L = len(ids)
        alive = (torch.arange(L, device=dev)[:, None, None] < ncl[None]) & inimg     # pixel still composites splat i
        hit = (alpha >= 1.0 / 255.0) & (power <= 0) & alive                          # [L,16,16]
        def blocks(bh, bw):
            v = hit.view(L, 16 // bh, bh, 16 // bw, bw).any(4).any(2)                # [L, nbh, nbw] block touched
            al = alive.view(L, 16 // bh, bh, 16 // bw, bw).any(4).any(2)             # block still has live pixels
            return int((v & al).sum()) * bh * bw, int((v & al).sum())
        # conservative bbox of the alpha >= 1/255 ellipse (what the kernel's culling uses)
        a_, b_, c_ = -2 * A, -B, -2 * Cc
        tau = torch.log(255.0 * op).clamp(min=0)
        det = (a_ * c_ - b_ * b_).clamp(min=1e-12)
        ex = torch.sqrt(2 * tau * c_ / det); ey = torch.sqrt(2 * tau * a_ / det)
        inb = ((px[None, None, :] >= (x - ex)[:, None, None]) & (px[None, None, :] <= (x + ex)[:, None, None]) &
               (py[None, :, None] >= (y - ey)[:, None, None]) & (py[None, :, None] <= (y + ey)[:, None, None]))
        for nm, (bh, bw) in (("bbhalf", (8, 16)), ("bbquad", (8, 8)), ("bb4x4", (4, 4))):
            v = inb.view(L, 16 // bh, bh, 16 // bw, bw).any(4).any(2)
            al = alive.view(L, 16 // bh, bh, 16 // bw, bw).any(4).any(2)
            tot[nm] = tot.get(nm, 0) + int((v & al).sum()) * bh * bw
        tot["tile"] += int(alive.view(L, -1).any(1).sum()) * 256
        tot["half"] += blocks(8, 16)[0]; tot["halfvis"] += blocks(8, 16)[1]
        tot["quad"] += blocks(8, 8)[0]; tot["quadvis"] += blocks(8, 8)[1]
        tot["row2"] += blocks(2, 16)[0]
        tot["q8x2"] += blocks(2, 8)[0]
        tot["q4x4"] += blocks(4, 4)[0]
        tot["q8x4"] = tot.get("q8x4", 0) + blocks(8, 4)[0]          # 8 rows x 4 columns (two pixels per lane, vertically)
        tot["q4x8"] = tot.get("q4x8", 0) + blocks(4, 8)[0]
        tot["exact"] += int(hit.sum())
        tot["lens"] += L
    print("sampled tiles:", len(tiles), "mean list length", tot["lens"] / len(tiles))
    for k in ("tile", "bbhalf", "half", "bbquad", "quad", "row2", "q8x2", "q8x4", "q4x8", "bb4x4", "q4x4", "exact"):
        print(f"{k:6s} pairs/tile {tot[k] / len(tiles):10.0f}   x exact {tot[k] / max(tot['exact'], 1):6.2f}")
    print("half-tile visits/tile", tot["halfvis"] / len(tiles), " quadrant visits/tile", tot["quadvis"] / len(tiles))
