"""Deformable_Field OFF the reference's operating point (any --d / --w, use_softmax, split_xyz [REF scene/deformable_field.py:74-127,
options/gaussian_option.py:54-55]): the layer-by-layer HIP path (csrc/deform_generic.hip, deform_ops.GenericMlp) against vectors the
reference's own class produced (tests/golden/make_golden_generic_mlp.py), against a float64 torch restatement at sizes the fixture does
not hold, and through GaussianModel / TrainStep.  The CPU part pins the parameter names (= state_dict keys) as data."""
import os

import numpy as np
import pytest
import torch

import gaussianprediction_amd as gpa
from util import rel_l2

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "deformable_field_generic.npz")
TAGS = ("a", "b", "c", "d", "e", "f")


def _net(g, tag):
    d_in, d_out, d, w, sm, split, M, seed = [int(v) for v in g[f"{tag}_meta"]]
    net = gpa.Deformable_Field(d_in, output_dim=d_out, d=d, w=w, use_softmax=bool(sm), split_xyz=bool(split))
    return net, (d_in, d_out, d, w, bool(sm), bool(split), M)


def test_generic_parameter_names_are_the_references():
    g = np.load(G)
    for tag in TAGS:
        net, meta = _net(g, tag)
        assert net.generic
        assert [k for k, _ in net.named_parameters()] == [str(k) for k in g[f"{tag}_names"]], tag
        sd = {str(k): torch.tensor(g[f"{tag}_grad_{k}"]) for k in g[f"{tag}_names"]}          # (any tensors of the right shapes)
        net.load_state_dict(sd)                                                                # a reference checkpoint's keys load
    assert not gpa.Deformable_Field(104, output_dim=7, d=4, w=256).generic                     # the operating point keeps the fused kernels
    with pytest.raises(ValueError):
        gpa.Deformable_Field(10, d=0)


def test_generic_path_has_no_cpu_fallback():
    net = gpa.Deformable_Field(12, output_dim=3, d=2, w=8)
    with pytest.raises(RuntimeError):
        net(torch.zeros(4, 12))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_generic_mlp_matches_reference_golden_vectors(tag):
    g = np.load(G)
    net, (d_in, d_out, d, w, sm, split, M) = _net(g, tag)
    net.load_state_dict({str(k): torch.tensor(v) for k, v in _params_of(g, tag).items()})
    net = net.cuda()
    x = torch.tensor(g[f"{tag}_x"], device="cuda", requires_grad=True)
    y = net(x)
    assert tuple(y.shape) == tuple(g[f"{tag}_y"].shape)
    (y * torch.tensor(g[f"{tag}_gy"], device="cuda")).sum().backward()
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[f"{tag}_y"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{tag}_dx"], rtol=1e-3, atol=2e-6)
    for k, p in net.named_parameters():
        want = g[f"{tag}_grad_{k}"]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()) + 1e-6, (tag, k)


def _params_of(g, tag):
    """the fixture's parameter VALUES, regenerated as its script made them (numpy PCG64, named_parameters() order)"""
    d_in, d_out, d, w, sm, split, M, seed = [int(v) for v in g[f"{tag}_meta"]]
    rng = np.random.default_rng(seed)
    out = {}
    for k in g[f"{tag}_names"]:
        shape = g[f"{tag}_grad_{k}"].shape
        fan_in = shape[1] if len(shape) == 2 else shape[0]
        out[str(k)] = rng.uniform(-1, 1, size=shape).astype(np.float32) / np.float32(np.sqrt(max(fan_in, 1)))
    return out


def _ref64(net, x64):
    """float64 restatement of the module on the CPU (plain torch)"""
    def chain(mlp, head):
        h = x64
        for i in range(net.d):
            h = torch.relu(h @ mlp[2 * i].weight.detach().double().cpu().T + mlp[2 * i].bias.detach().double().cpu())
        o = h @ head[0].weight.detach().double().cpu().T + head[0].bias.detach().double().cpu()
        return torch.softmax(o, -1) if net.use_softmax else o
    if net.split_xyz:
        return torch.cat([chain(net.mlp[f"mlp{k}"], net.feature_to_deformation[f"feature_to_deformation{k}"]) for k in range(net.output_times)], -1)
    return chain(net.mlp, net.feature_to_deformation)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,d,w,sm,split", [(1, 1, 1, False, False), (63, 2, 65, False, False), (5000, 8, 256, False, False), (20003, 3, 128, True, False),
                                               (4097, 2, 40, False, True), (300, 5, 512, False, False)])
def test_generic_mlp_against_float64(rows, d, w, sm, split):
    torch.manual_seed(rows + d)
    d_in, d_out = 37, 6
    net = gpa.Deformable_Field(d_in, output_dim=d_out, d=d, w=w, use_softmax=sm, split_xyz=split).cuda()
    x = (torch.rand(rows, d_in, device="cuda") * 2 - 1).requires_grad_(True)
    y = net(x)
    gy = torch.randn_like(y)
    (y * gy).sum().backward()
    x64 = x.detach().double().cpu().requires_grad_(True)
    ps = list(net.parameters())
    y64 = _ref64(net, x64)
    # (autograd through the restatement: the parameters enter as constants there, so their reference gradients come from a second pass)
    (y64 * gy.double().cpu()).sum().backward()
    assert (y.detach().double().cpu() - y64.detach()).abs().max() < 2e-5 * max(1.0, float(y64.detach().abs().max()))
    assert rel_l2(x.grad.cpu().numpy(), x64.grad.numpy()) < 1e-4
    ref = gpa.Deformable_Field(d_in, output_dim=d_out, d=d, w=w, use_softmax=sm, split_xyz=split).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
    ref.generic = True
    torch_y = _torch_forward(ref, x.detach().double().cpu())
    (torch_y * gy.double().cpu()).sum().backward()
    for (k, p), q in zip(net.named_parameters(), ref.parameters()):
        want = q.grad if q.grad is not None else torch.zeros_like(q)
        got = p.grad.double().cpu() if p.grad is not None else torch.zeros_like(want)
        assert float((got - want).norm()) <= 1e-4 * float(want.norm()) + 1e-6 * rows ** 0.5, (k, float((got - want).norm()), float(want.norm()))
    assert len(ps) == (2 * (d + 1)) * (d_out if split else 1)


def _torch_forward(net, x):
    """the module's own Parameters through plain torch ops (float64, CPU): the reference for the weight gradients"""
    def chain(mlp, head):
        h = x
        for i in range(net.d):
            h = torch.relu(torch.nn.functional.linear(h, mlp[2 * i].weight, mlp[2 * i].bias))
        o = torch.nn.functional.linear(h, head[0].weight, head[0].bias)
        return torch.softmax(o, -1) if net.use_softmax else o
    if net.split_xyz:
        return torch.cat([chain(net.mlp[f"mlp{k}"], net.feature_to_deformation[f"feature_to_deformation{k}"]) for k in range(net.output_times)], -1)
    return chain(net.mlp, net.feature_to_deformation)


@pytest.mark.gpu
def test_generic_forward_fused_equals_the_fused_kernels_encoding():
    """forward_fused builds [feature | PE(xyz) | PE(t)] with gp_mlp_input_forward: the same input, element for element, as the fused
    kernels build in LDS -- checked by giving BOTH paths the operating-point network (the generic one forced) and comparing outputs
    and input gradients."""
    torch.manual_seed(3)
    rows, F = 3000, 8
    net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256).cuda()
    feat = (torch.rand(rows, 32, device="cuda") - 0.5).requires_grad_(True)
    xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3).requires_grad_(True)
    t = torch.tensor([0.37], device="cuda")
    gy = torch.randn(rows, 7, device="cuda")
    outs = []
    for generic in (False, True):
        net.generic = generic
        for p in list(net.parameters()) + [feat, xyz]:
            p.grad = None
        y = net.forward_fused(feat, xyz, t, 10, F)
        (y * gy).sum().backward()
        outs.append((y.detach().clone(), feat.grad.clone(), xyz.grad.clone(), [p.grad.clone() for p in net.parameters()]))
    (ya, fa, xa, pa), (yb, fb, xb, pb) = outs
    assert (ya - yb).abs().max() < 2e-5
    assert rel_l2(fa.cpu().numpy(), fb.cpu().numpy()) < 1e-4 and rel_l2(xa.cpu().numpy(), xb.cpu().numpy()) < 1e-4
    for a, b in zip(pa, pb):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-4


@pytest.mark.gpu
def test_model_with_another_depth_and_width_renders_and_trains():
    """--d 2 --w 64 end to end: GaussianModel builds the generic network, render() and a TrainStep step (graph path: the one-call step
    is for the operating point) run, the loss falls, and the per-frame deformation equals the float64 restatement's."""
    from test_gpu_render import build, make_args
    from gaussianprediction_amd.train_step import TrainStep
    pc, cam, P, sd, raw, raw_w, idx, args = build(N=1500, K=40, W=96, H=80, args=make_args(d=2, w=64))
    assert pc.df_model.generic and pc.df_model.d == 2 and pc.df_model.w == 64
    gt = torch.rand(3, 80, 96, generator=torch.Generator().manual_seed(1)).cuda()
    ts = TrainStep(pc, [cam], [gt], 50000, speculative=True)
    losses = [float(ts.step(i)[0]) for i in range(12)]
    torch.cuda.synchronize()
    assert ts.fused_steps == 0
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    for p in pc.parameters():
        assert torch.isfinite(p).all()
