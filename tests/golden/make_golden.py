#!/usr/bin/env python
"""Generates tests/golden/*.npz by RUNNING the importable pieces of the reference in this container.

Run here only (needs /root/reference; never runs on the GPU box):   python tests/golden/make_golden.py
Only data (inputs + expected outputs) is written -- no reference source text.

Pieces of the reference that import and run on CPU:
  scene/deformable_field.py   positional_encoding, Deformable_Field     (loaded by file path)
  utils/sh_utils.py           eval_sh
  utils/graphics_utils.py     getWorld2View2, getProjectionMatrix
  utils/loss_utils.py         l1_loss, ssim
  utils/general_utils.py      build_rotation/build_scaling_rotation/strip_symmetric (hard-code
                              device="cuda": executed from source with the device string replaced)
  utils/camera_utils.py       quat_mul (module import fails on pytorch3d: the one function is
                              extracted with `ast` and executed)
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def numpy_linear_weights(rng, out_f, in_f):
    """Deterministic weights (numpy PCG64, stable across versions) shaped like nn.Linear's."""
    bound = 1.0 / np.sqrt(in_f)
    w = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    return w, b


def mlp_state(seed, d_in, d_out, d=4, w=256):
    rng = np.random.default_rng(seed)
    sd = {}
    for i in range(d):
        W, b = numpy_linear_weights(rng, w, d_in if i == 0 else w)
        sd[f"mlp.{2 * i}.weight"], sd[f"mlp.{2 * i}.bias"] = W, b
    W, b = numpy_linear_weights(rng, d_out, w)
    sd["feature_to_deformation.0.weight"], sd["feature_to_deformation.0.bias"] = W, b
    return sd


def main():
    torch.manual_seed(0)
    sys.path.insert(0, REF)
    df = load_by_path("ref_deformable_field", "scene/deformable_field.py")
    sh = load_by_path("ref_sh_utils", "utils/sh_utils.py")
    gu = load_by_path("ref_graphics_utils", "utils/graphics_utils.py")
    lu = load_by_path("ref_loss_utils", "utils/loss_utils.py")

    # ---- positional encoding ----------------------------------------------------------------
    rng = np.random.default_rng(1)
    xyz = (rng.uniform(-1.5, 1.5, size=(64, 3))).astype(np.float32)
    pe = {"xyz": xyz, "xyz_pe10": df.positional_encoding(torch.tensor(xyz), 10).numpy()}
    for F in (6, 8, 10):
        for t in (0.0, 0.3, 1.0):
            pe[f"t{t}_F{F}"] = df.positional_encoding(torch.tensor([t], dtype=torch.float32), F).numpy()
    np.savez_compressed(os.path.join(OUT, "posenc.npz"), **pe)

    # ---- Deformable_Field forward + autograd ---------------------------------------------------
    out = {}
    for tag, (d_in, d_out, M, seed) in {"a": (104, 7, 257, 11), "b": (112, 8, 7, 12), "c": (108, 7, 1, 13)}.items():
        net = df.Deformable_Field(d_in, output_dim=d_out, d=4, w=256, split_xyz=False)
        sd = mlp_state(seed, d_in, d_out)
        net.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
        x = torch.tensor(np.random.default_rng(seed + 100).uniform(-1, 1, size=(M, d_in)).astype(np.float32), requires_grad=True)
        y = net(x)
        gy = torch.tensor(np.random.default_rng(seed + 200).normal(size=(M, d_out)).astype(np.float32))
        (y * gy).sum().backward()
        out[f"{tag}_meta"] = np.array([d_in, d_out, M, seed])
        out[f"{tag}_y"] = y.detach().numpy()
        out[f"{tag}_dx"] = x.grad.numpy()
        for k, p in net.named_parameters():
            g = p.grad.numpy()
            if tag == "a":
                out[f"{tag}_grad_{k}"] = g
            else:
                out[f"{tag}_gradsum_{k}"] = np.array([g.sum(dtype=np.float64), np.abs(g).sum(dtype=np.float64)])
    np.savez_compressed(os.path.join(OUT, "deformable_field.npz"), **out)

    # ---- SH -> RGB (eval_sh + 0.5, clamp) as gaussian_renderer/__init__.py:86-91 ----------------
    rng = np.random.default_rng(2)
    N = 128
    feats = rng.normal(size=(N, 16, 3)).astype(np.float32)          # get_features layout [N,16,3]
    pts = rng.uniform(-2, 2, size=(N, 3)).astype(np.float32)
    campos = np.array([0.3, -0.2, 4.0], np.float32)
    d = torch.tensor(pts) - torch.tensor(campos)[None]
    d = d / d.norm(dim=1, keepdim=True)
    shd = {"features": feats, "points": pts, "campos": campos}
    shs_view = torch.tensor(feats).transpose(1, 2).reshape(-1, 3, 16)
    for deg in range(4):
        rgb = torch.clamp_min(sh.eval_sh(deg, shs_view, d) + 0.5, 0.0)
        shd[f"rgb_deg{deg}"] = rgb.numpy()
    np.savez_compressed(os.path.join(OUT, "sh.npz"), **shd)

    # ---- cov3D from scale/quaternion (general_utils, device string substituted) ----------------
    src = open(os.path.join(REF, "utils/general_utils.py")).read().replace('"cuda"', '"cpu"').replace("'cuda'", "'cpu'")
    ns = {}
    exec(compile(src, "general_utils_cpu", "exec"), ns)
    rng = np.random.default_rng(3)
    s = np.exp(rng.uniform(-4, -1, size=(96, 3))).astype(np.float32)
    q = rng.normal(size=(96, 4)).astype(np.float32)
    L = ns["build_scaling_rotation"](1.0 * torch.tensor(s), torch.tensor(q))
    cov = ns["strip_symmetric"](L @ L.transpose(1, 2))
    Rm = ns["build_rotation"](torch.tensor(q))
    np.savez_compressed(os.path.join(OUT, "cov3d.npz"), scales=s, quats=q, cov3D=cov.numpy(), R=Rm.numpy())

    # ---- quat_mul ------------------------------------------------------------------------------
    tree = ast.parse(open(os.path.join(REF, "utils/camera_utils.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "quat_mul"][0]
    ns2 = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "quat_mul_only", "exec"), ns2)
    q1 = rng.normal(size=(64, 4)).astype(np.float32)
    q2 = rng.normal(size=(64, 4)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "quat_mul.npz"), q1=q1, q2=q2,
                        q1q2=ns2["quat_mul"](torch.tensor(q1), torch.tensor(q2)).numpy())

    # ---- camera matrices (scene/cameras.py:59-62 restated with the reference's own helpers) -----
    cams = {}
    rng = np.random.default_rng(4)
    for i in range(6):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        T = rng.uniform(-3, 3, size=3)
        fovx, fovy = rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.2)
        trans = np.array([0.0, 0.0, 0.0]) if i < 4 else rng.uniform(-1, 1, size=3)
        scale = 1.0 if i < 4 else 1.5
        wv = torch.tensor(gu.getWorld2View2(Q, T, trans, scale)).transpose(0, 1)
        pj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wv.unsqueeze(0).bmm(pj.unsqueeze(0))).squeeze(0)
        cams[f"{i}_R"], cams[f"{i}_T"], cams[f"{i}_fov"] = Q, T, np.array([fovx, fovy])
        cams[f"{i}_trans"], cams[f"{i}_scale"] = trans, np.array(scale)
        cams[f"{i}_view"], cams[f"{i}_proj"], cams[f"{i}_full"] = wv.numpy(), pj.numpy(), full.numpy()
        cams[f"{i}_center"] = wv.inverse()[3, :3].numpy()
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **cams)

    # ---- loss: 0.8 L1 + 0.2 (1 - SSIM) and dL/dimage (train.py:105-108) -------------------------
    rng = np.random.default_rng(5)
    img = torch.tensor(rng.uniform(0, 1, size=(3, 45, 52)).astype(np.float32), requires_grad=True)
    gt = torch.tensor(rng.uniform(0, 1, size=(3, 45, 52)).astype(np.float32))
    l1 = lu.l1_loss(img, gt)
    ss = lu.ssim(img, gt)
    loss = 0.8 * l1 + 0.2 * (1.0 - ss)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), img=img.detach().numpy(), gt=gt.numpy(), l1=l1.item(),
                        ssim=ss.item(), loss=loss.item(), dimg=img.grad.numpy())
    # ---- optimisation defaults + learning-rate schedules (arguments/__init__.py:72-99, utils/general_utils.py:29-62, the five
    #      schedules of scene/gaussian_model.py training_setup) ------------------------------------------------------------
    import argparse
    ar = load_by_path("ref_arguments", "arguments/__init__.py")
    op = ar.OptimizationParams(argparse.ArgumentParser())
    defaults = {k: float(v) for k, v in vars(op).items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
    steps = np.array([0, 1, 10, 100, 500, 1000, 5000, 15000, 29999, 30000, 30001, 45000, 60000, 100000], dtype=np.int64)
    sched = {}
    for name, kw in dict(
            xyz=dict(lr_init=op.position_lr_init * 5.0, lr_final=op.position_lr_final * 5.0, lr_delay_mult=op.position_lr_delay_mult,
                     max_steps=op.position_lr_max_steps),
            mlp=dict(lr_init=op.mlp_lr, lr_final=op.position_lr_final, lr_delay_mult=op.position_lr_delay_mult,
                     max_steps=op.position_lr_max_steps),
            hash=dict(lr_init=op.hash_lr, lr_final=op.hash_lr_final, lr_delay_steps=op.position_lr_max_steps,
                      lr_delay_mult=op.position_lr_delay_mult, max_steps=op.position_lr_max_steps),
            mfeature=dict(lr_init=op.mfeature_lr, lr_final=op.mfeature_lr_final, lr_delay_mult=op.position_lr_delay_mult,
                          max_steps=op.position_lr_max_steps),
            plain=dict(lr_init=1e-2, lr_final=1e-4, max_steps=2000), off=dict(lr_init=0.0, lr_final=0.0)).items():
        f = ns["get_expon_lr_func"](**kw)
        sched[name] = np.array([f(int(t)) for t in steps], dtype=np.float64)
        sched[name + "_kw"] = np.array([kw.get("lr_init"), kw.get("lr_final"), kw.get("lr_delay_steps", 0), kw.get("lr_delay_mult", 1.0),
                                        kw.get("max_steps", 1000000)], dtype=np.float64)
    xs = np.array([1e-4, 0.01, 0.1, 0.5, 0.9, 0.999], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "training.npz"), steps=steps, default_names=np.array(sorted(defaults)),
                        default_values=np.array([defaults[k] for k in sorted(defaults)]), inv_sig_x=xs,
                        inv_sig_y=ns["inverse_sigmoid"](torch.tensor(xs)).numpy(), **sched)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
