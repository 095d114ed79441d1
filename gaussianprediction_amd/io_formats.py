"""On-disk formats of the reference, so that scenes trained with either code base load
in the other:

* `point_cloud.ply` as written by `GaussianModel.save_ply` [REF scene/gaussian_model.py:493-524]: binary little-endian,
  one `vertex` element, float32 columns
  `x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3`, where f_dc / f_rest are the [N,1,3] / [N,15,3]
  tensors transposed to channel-major ([N,3,1] / [N,3,15]) and flattened, normals are zero, and every value is the RAW
  parameter (log-scale, logit-opacity, un-normalised quaternion).  (The reference writes it through `plyfile`, which is
  not installed here: the header below is the one plyfile emits for that dtype list.)
* `chkpnt<iteration>.pth` = `torch.save((gaussians.state_dict(), optimizer.state_dict(), iteration))`
  [REF train.py:199-201], restored with `load_state_dict(strict=False)` after sizing the model from `_xyz` /
  `super_gaussians` [REF train.py:48-57].  Parameter names are identical here, so the tuple is interchangeable.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def ply_attributes(n_dc=3, n_rest=45, n_scale=3, n_rot=4):
    """[REF scene/gaussian_model.py:493-506 construct_list_of_attributes]"""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names += ["opacity"]
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(model, path):
    """Write `model`'s Gaussians exactly as the reference's save_ply does."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if hasattr(model, "_sync_side_stream"):
        model._sync_side_stream()   # a training harness may still be updating / gathering parameters (second stream, all-gather)
    xyz = model._xyz.detach().cpu().numpy()
    normals = np.zeros_like(xyz)
    f_dc = model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    opac = model._opacity.detach().cpu().numpy()
    scale = model._scaling.detach().cpu().numpy()
    rot = model._rotation.detach().cpu().numpy()
    table = np.concatenate((xyz, normals, f_dc, f_rest, opac, scale, rot), axis=1).astype("<f4")
    names = ply_attributes(f_dc.shape[1], f_rest.shape[1], scale.shape[1], rot.shape[1])
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {table.shape[0]}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(table).tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4", "int32": "<i4",
              "uint": "<u4", "uint32": "<u4"}


def read_ply_vertices(path):
    """Structured numpy array of the `vertex` element of a binary-little-endian or ascii PLY (scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties on the vertex element are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        dt = np.dtype(props)
        if fmt == "binary_little_endian":
            return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            out = np.empty(count, dtype=dt)
            for k, (name, _) in enumerate(props):
                out[name] = rows[:, k]
            return out
        raise ValueError(f"{path}: unsupported PLY format {fmt}")


def load_ply(path, sh_degree=3, device="cpu"):
    """Raw parameter tensors from a reference-format point_cloud.ply (inverse of save_ply)."""
    v = read_ply_vertices(path)
    n_rest = 3 * ((sh_degree + 1) ** 2 - 1)
    cols = lambda names: np.stack([np.asarray(v[n], dtype=np.float32) for n in names], axis=1)   # noqa: E731
    xyz = cols(["x", "y", "z"])
    f_dc = cols([f"f_dc_{i}" for i in range(3)]).reshape(-1, 3, 1)
    rest_names = [n for n in v.dtype.names if n.startswith("f_rest_")]
    if len(rest_names) != n_rest:
        raise ValueError(f"{path}: {len(rest_names)} f_rest columns, sh_degree {sh_degree} needs {n_rest}")
    f_rest = cols([f"f_rest_{i}" for i in range(n_rest)]).reshape(-1, 3, n_rest // 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)                               # noqa: E731
    return dict(xyz=t(xyz), features_dc=t(f_dc).transpose(1, 2).contiguous(), features_rest=t(f_rest).transpose(1, 2).contiguous(),
                opacity=t(cols(["opacity"])), scaling=t(cols([n for n in v.dtype.names if n.startswith("scale_")])),
                rotation=t(cols([n for n in v.dtype.names if n.startswith("rot_")])))


def save_checkpoint(model, optimizer_state, iteration, path):
    """[REF train.py:199-201]"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if hasattr(model, "_sync_side_stream"):
        model._sync_side_stream()
    torch.save((model.state_dict(), optimizer_state, iteration), path)


def load_checkpoint(path, args, sh_degree=3, time_input_dim=None, xyz_input_dim=60, device="cpu"):
    """Rebuild a GaussianModel from a reference-format checkpoint [REF train.py:48-57].  Returns
    (model, optimizer_state, iteration)."""
    from .gaussian_model import GaussianModel
    model_params, opt_state, iteration = torch.load(path, map_location=device, weights_only=False)
    if time_input_dim is None:                       # D_in = feature_dim + xyz_input_dim + time_input_dim
        time_input_dim = model_params["df_model.mlp.0.weight"].shape[1] - args.feature_dim - xyz_input_dim
    m = GaussianModel(sh_degree, args)
    m.set_inputDim(time_input_dim, xyz_input_dim)
    kp = model_params.get("super_gaussians")
    m.create_from_tensors(model_params["_xyz"], model_params["_features_dc"], model_params["_features_rest"],
                          model_params["_scaling"], model_params["_rotation"], model_params["_opacity"],
                          model_params["motion_feature"], kp, model_params.get("super_gaussians_feature"),
                          with_weights_model="weights_model.params" in model_params)
    missing, unexpected = m.load_state_dict(model_params, strict=False)
    m.active_sh_degree = m.max_sh_degree
    return m.to(device), opt_state, iteration
