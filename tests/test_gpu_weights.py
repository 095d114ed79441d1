"""GPU parity of the keypoint-weights producers against oracle/weights_oracle.py.
parity unpinned: the oracle restates the published tcnn / frnn algorithms, not the absent dependencies themselves."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gaussianprediction_amd as gpa  # noqa: E402
from gaussianprediction_amd.weights_ops import WeightsModel, knn_keypoints  # noqa: E402
from oracle import weights_oracle as wo  # noqa: E402
from host_checkers import weights_model_unfused  # noqa: E402


def _small_model(n_out=12, log2_T=12, levels=16):
    return WeightsModel(n_out, n_levels=levels, log2_hashmap_size=log2_T, base_resolution=16, seed=7, device="cuda")


@pytest.mark.parametrize("n", [0, 1, 777])
def test_hashgrid_and_mlp_forward_matches_oracle(n):
    m = _small_model()
    meta = wo.grid_meta(16, 4, 12, 16)
    assert meta["total"] == m.table_entries
    rng = np.random.default_rng(n)
    xyz = torch.tensor(rng.uniform(-1.7, 1.7, size=(n, 3)).astype(np.float32))     # includes negative cells (uint32 wrap)
    with torch.no_grad():
        # make the table non-trivial
        m.params[wo_mlp():] = torch.tensor(rng.normal(size=m.params.numel() - wo_mlp()).astype(np.float32)).cuda()
    out = m(xyz.cuda())
    ref = wo.weights_model(xyz.double(), m.params.detach().cpu().double(), meta, 12)
    assert out.shape == (n, 12)
    if n:
        # the encoding itself is exact in float32 up to the blend's summation; the MLP goes through library GEMMs
        assert np.abs(out.detach().cpu().numpy() - ref.numpy()).max() < 2e-4 * max(1.0, float(ref.abs().max()))


def wo_mlp():
    return 64 * 64 + 64 * 64 + 16 * 64


def test_hashgrid_encoding_bit_exact():
    from gaussianprediction_amd.weights_ops import _HashGridEncode
    m = _small_model(log2_T=10)
    meta = wo.grid_meta(16, 4, 10, 16)
    rng = np.random.default_rng(3)
    xyz = torch.tensor(rng.uniform(-3, 3, size=(513, 3)).astype(np.float32))
    table = torch.tensor(rng.normal(size=(m.table_entries, 4)).astype(np.float32))
    enc = _HashGridEncode.apply(xyz.cuda(), table.cuda(), m.cfg).cpu().numpy()
    ref = wo.hash_encode(xyz, table, meta).numpy()
    assert np.array_equal(enc, ref)          # same float32 expression tree: identical bits


def test_spatial_order_changes_nothing():
    """perm only decides which points share a wavefront: forward identical, table gradient equal up to atomic order."""
    from gaussianprediction_amd.weights_ops import _HashGridEncode, morton_order
    m = _small_model(log2_T=11)
    rng = np.random.default_rng(11)
    xyz = torch.tensor(rng.uniform(-1.5, 1.5, size=(5000, 3)).astype(np.float32)).cuda()
    table = torch.tensor(rng.normal(size=(m.table_entries, 4)).astype(np.float32)).cuda()
    gy = torch.tensor(rng.normal(size=(5000, 64)).astype(np.float32)).cuda()
    perm = morton_order(xyz)
    assert sorted(perm.cpu().tolist()) == list(range(5000))
    outs, grads = [], []
    for pm in (None, perm):
        t = table.clone().requires_grad_(True)
        o = _HashGridEncode.apply(xyz, t, m.cfg, pm)
        (o * gy).sum().backward()
        outs.append(o.detach().cpu().numpy()); grads.append(t.grad.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.linalg.norm(grads[0] - grads[1]) <= 1e-5 * np.linalg.norm(grads[0])


def test_fused_matches_unfused_with_spatial_order():
    """n > 4096: Morton order + persistent fused kernels vs the encode kernel + library GEMMs."""
    m = _small_model(log2_T=13)
    rng = np.random.default_rng(21)
    xyz = torch.tensor(rng.uniform(-1.5, 1.5, size=(9001, 3)).astype(np.float32)).cuda()
    gy = torch.tensor(rng.normal(size=(9001, 12)).astype(np.float32)).cuda()
    with torch.no_grad():
        m.params[wo_mlp():] = torch.tensor(rng.normal(size=m.params.numel() - wo_mlp()).astype(np.float32)).cuda()
    res = []
    for fused in (True, False):
        m.params.grad = None
        o = m(xyz) if fused else weights_model_unfused(m, xyz, m.spatial_order(xyz))
        (o * gy).sum().backward()
        res.append((o.detach().cpu().numpy(), m.params.grad.cpu().numpy()))
    assert np.abs(res[0][0] - res[1][0]).max() < 2e-4 * max(1.0, np.abs(res[1][0]).max())
    rel = np.linalg.norm(res[0][1] - res[1][1]) / np.linalg.norm(res[1][1])
    assert rel < 1e-4, rel


@pytest.mark.parametrize("fused", [True, False])
def test_hashgrid_backward_matches_autograd(fused):
    m = _small_model(log2_T=9)
    meta = wo.grid_meta(16, 4, 9, 16)
    rng = np.random.default_rng(5)
    xyz = torch.tensor(rng.uniform(-1.2, 1.2, size=(300, 3)).astype(np.float32))
    gy = torch.tensor(rng.normal(size=(300, 12)).astype(np.float32))
    out = m(xyz.cuda()) if fused else weights_model_unfused(m, xyz.cuda())
    (out * gy.cuda()).sum().backward()
    p64 = m.params.detach().cpu().double().requires_grad_(True)
    ref = wo.weights_model(xyz.double(), p64, meta, 12)
    (ref * gy.double()).sum().backward()
    g, gr = m.params.grad.cpu().numpy(), p64.grad.numpy()
    rel = np.linalg.norm(g - gr) / max(np.linalg.norm(gr), 1e-30)
    assert rel < 1e-5, rel
    # rows of the padded output layer beyond n_out get no gradient
    assert np.all(g[8192 + 12 * 64: wo_mlp()] == 0)


@pytest.mark.parametrize("knn_type,K,nn", [("3D", 100, 6), ("hybird", 250, 6), ("hybird", 512, 8), ("3D", 6, 6)])
def test_knn_matches_oracle(knn_type, K, nn):
    rng = np.random.default_rng(K + nn)
    N = 3000
    xyz = rng.uniform(-1.3, 1.3, size=(N, 3)).astype(np.float32)
    feat = (1e-1 * rng.uniform(-1, 1, size=(N, 32))).astype(np.float32)
    kp = xyz[rng.choice(N, K, replace=False)] + 0.01 * rng.normal(size=(K, 3)).astype(np.float32)
    kpf = (1e-1 * rng.uniform(-1, 1, size=(K, 32))).astype(np.float32)
    idx, d2 = knn_keypoints(torch.tensor(xyz).cuda(), torch.tensor(kp).cuda(), nn, torch.tensor(feat).cuda(),
                            torch.tensor(kpf).cuda(), 5.0, knn_type, return_dist=True)
    if knn_type == "3D":
        ri, rd = wo.knn(xyz, kp, nn)
    else:
        ri, rd = wo.knn(np.concatenate([xyz, np.float32(5.0) * feat], 1), np.concatenate([kp, np.float32(5.0) * kpf], 1), nn)
    assert np.array_equal(idx.cpu().numpy(), ri)          # index work: bit-exact
    assert np.array_equal(d2.cpu().numpy(), rd)


@pytest.mark.parametrize("feat_scale", [1e-3, 1e-1, 1.0])
def test_knn_visiting_order_does_not_change_the_result(feat_scale):
    """`order` regroups the points into spatially coherent wavefronts so that a wavefront can drop a keypoint after a prefix of
    the dimensions; indices AND distances stay those of the plain kernel (and of the oracle) whatever the order is."""
    from gaussianprediction_amd.weights_ops import morton_order
    rng = np.random.default_rng(7)
    N, K, nn = 40000, 250, 6
    xyz = torch.tensor(rng.uniform(-1.3, 1.3, size=(N, 3)).astype(np.float32)).cuda()
    feat = torch.tensor((feat_scale * rng.uniform(-1, 1, size=(N, 32))).astype(np.float32)).cuda()
    kp = xyz[torch.tensor(rng.choice(N, K, replace=False)).cuda()].clone()
    kp[:10] = kp[10:20]                                     # coincident keypoints: ties go to the lower index
    kpf = feat[:K].clone()
    kpf[:10] = kpf[10:20]
    base_i, base_d = knn_keypoints(xyz, kp, nn, feat, kpf, 5.0, "hybird", return_dist=True)
    for order in (morton_order(xyz), torch.randperm(N, generator=torch.Generator().manual_seed(1)).to(torch.int32).cuda()):
        i2, d2 = knn_keypoints(xyz, kp, nn, feat, kpf, 5.0, "hybird", return_dist=True, order=order)
        assert torch.equal(i2, base_i) and torch.equal(d2, base_d)
    sub = slice(0, 2000)
    ri, rd = wo.knn(np.concatenate([xyz.cpu().numpy(), np.float32(5.0) * feat.cpu().numpy()], 1)[sub],
                    np.concatenate([kp.cpu().numpy(), np.float32(5.0) * kpf.cpu().numpy()], 1), nn)
    assert np.array_equal(base_i[sub].cpu().numpy(), ri) and np.array_equal(base_d[sub].cpu().numpy(), rd)
    with pytest.raises(RuntimeError):
        knn_keypoints(xyz, kp, nn, feat, kpf, 5.0, "hybird", order=torch.arange(N, device="cuda"))    # int64: refused


def test_model_forward_computes_its_own_weights():
    """Stage-3 forward without set_keypoint_weights: weights model + kNN per frame, as the reference [REF :257-262]."""
    from types import SimpleNamespace
    from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians, make_keypoints
    margs = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                            jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=False,
                            step_opacity_iteration=5000, opacity_type="implicit", xyz_noise_iteration=0, knn_type="hybird",
                            feature_amplify=5.0)
    raw = make_gaussians(SceneSpec(n_gaussians=2000, extent=(1.3, 1.3, 1.3), scale_lo=0.01, scale_hi=0.05, seed=3), device="cuda")
    kp, kpf, idx_ref, _ = make_keypoints(raw["xyz"], raw["motion_feature"], 64, 6)
    pc = gpa.GaussianModel(3, margs)
    pc.set_inputDim(12, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"], kp, kpf, with_weights_model=True)
    t = torch.tensor([0.3], device="cuda")
    xyz_t, q_t, s, o = pc(t, 50000)
    assert torch.isfinite(xyz_t).all() and torch.isfinite(q_t).all()
    assert pc.nearest_mask.shape == (2000, 6)
    (xyz_t.sum() + q_t.sum()).backward()
    g = pc.weights_model.params.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
