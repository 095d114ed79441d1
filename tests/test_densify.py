"""Densify / prune with optimizer-state surgery [REF scene/gaussian_model.py:532-690, 739-760]."""
from types import SimpleNamespace

import numpy as np
import torch

import gaussianprediction_amd as gpa
from gaussianprediction_amd import densify as dn
from gaussianprediction_amd.cameras import orbit_cameras
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians
from gaussianprediction_amd.train_step import TrainStep


def _setup(n=40, iteration=5000):
    margs = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                            jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=False,
                            step_opacity_iteration=5000, opacity_type="implicit", xyz_noise_iteration=0)
    raw = make_gaussians(SceneSpec(n_gaussians=n, extent=(1.3, 1.3, 1.3), scale_lo=0.01, scale_hi=0.2, seed=5))
    pc = gpa.GaussianModel(3, margs)
    pc.set_inputDim(12, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"])
    cams = orbit_cameras(2, 4.0, 0.69, 32, 32)
    ts = TrainStep(pc, cams, [torch.zeros(3, 32, 32)] * 2, iteration)       # stage 1 (no kernels are launched in this test)
    return pc, ts


def test_clone_split_prune_and_moment_surgery():
    pc, ts = _setup()
    n = pc._xyz.shape[0]
    mom = ts.adam_moments()
    for k, p in enumerate([pc._xyz, pc._scaling, pc._features_rest, pc.motion_feature]):
        m, v = mom[id(p)]
        m.copy_(torch.arange(m.numel(), dtype=torch.float32).reshape(m.shape) + 1000 * k)
        v.copy_(2 * m)
    ts.optimizer.step_count = 17
    old = {k: v.detach().clone() for k, v in dn._per_gaussian(pc).items()}
    old_m = mom[id(pc._xyz)][0].clone()
    stats = dn.DensificationStats(n, "cpu")
    stats.denom += 1
    hot = torch.zeros(n, dtype=torch.bool); hot[[1, 4, 7, 20, 21]] = True
    stats.xyz_gradient_accum[hot] = 1.0                       # mean view-space gradient 1.0 >> threshold
    extent = 5.0
    with torch.no_grad():
        pc._opacity[30:33] = -10.0                            # sigmoid -> ~4.5e-5 < min_opacity: pruned
        pc._scaling[[1, 4]] = np.log(0.2)                     # > percent_dense*extent = 0.05: split
        pc._scaling[[7, 20, 21]] = np.log(0.01)               # small: cloned
        pc._opacity[[1, 4, 7, 20, 21]] = 2.0
    old = {k: v.detach().clone() for k, v in dn._per_gaussian(pc).items()}
    g = torch.Generator().manual_seed(0)
    n_clone, n_src, n_pruned = dn.densify_and_prune(pc, ts, stats, max_grad=0.0002, min_opacity=0.005, extent=extent,
                                                    max_screen_size=None, generator=g)
    assert (n_clone, n_src, n_pruned) == (3, 2, 3)
    N1 = n - 2 - 3 + 3 + 4                                    # - split sources - transparent + clones + 2 children per source
    for k, p in dn._per_gaussian(pc).items():
        assert p.shape[0] == N1, k
    survivors = torch.ones(n, dtype=torch.bool); survivors[[1, 4, 30, 31, 32]] = False
    assert torch.equal(pc._xyz.detach()[:n - 5], old["xyz"][survivors])
    # clones are exact copies of their sources, appended after the survivors [REF :668-688]
    assert torch.equal(pc._xyz.detach()[n - 5:n - 2], old["xyz"][[7, 20, 21]])
    assert torch.equal(pc._features_rest.detach()[n - 5:n - 2], old["f_rest"][[7, 20, 21]])
    # split children: scale / (0.8 N) in log space, other attributes repeated [REF :645-662]
    ch = pc._scaling.detach()[n - 2:]
    assert torch.allclose(ch, (old["scaling"][[1, 4]].exp() / 1.6).log().repeat(2, 1))
    assert torch.equal(pc._rotation.detach()[n - 2:], old["rotation"][[1, 4]].repeat(2, 1))
    assert not torch.equal(pc._xyz.detach()[n - 2:], old["xyz"][[1, 4]].repeat(2, 1))      # sampled around the source
    # Adam moments: survivors carried, new rows zero, step count kept; gradients are views of the new bucket
    m, v = ts.adam_moments()[id(pc._xyz)]
    assert torch.equal(m[:n - 5], old_m[survivors]) and float(m[n - 5:].abs().sum()) == 0.0 and torch.equal(v[:n - 5], 2 * m[:n - 5])
    assert ts.optimizer.step_count == 17
    assert pc._xyz.grad is not None and pc._xyz.grad.shape == pc._xyz.shape and pc._xyz.grad.data_ptr() >= ts.bucket.flat.data_ptr()
    # the MLP kept its parameters and moments
    assert any(id(p) in ts.adam_moments() for p in pc.df_model.parameters())
    assert stats.denom.shape[0] == N1 and float(stats.denom.sum()) == 0.0


def test_stats_and_reset_opacity():
    pc, ts = _setup()
    n = pc._xyz.shape[0]
    stats = dn.DensificationStats(n, "cpu")
    vs = torch.zeros(n, 3, requires_grad=True)
    vs.grad = torch.zeros(n, 3); vs.grad[:, 0] = 3.0; vs.grad[:, 1] = 4.0
    filt = torch.zeros(n, dtype=torch.bool); filt[::2] = True
    radii = torch.arange(n, dtype=torch.int32)
    stats.add(vs, filt, radii); stats.add(vs, filt, radii)
    assert torch.allclose(stats.xyz_gradient_accum[::2], torch.full((n // 2, 1), 10.0)) and float(stats.xyz_gradient_accum[1::2].sum()) == 0
    assert float(stats.denom[0]) == 2 and float(stats.max_radii2D[2]) == 2.0
    mom = ts.adam_moments()[id(pc._opacity)]
    mom[0].fill_(1.0)
    dn.reset_opacity(pc, ts)
    assert float(torch.sigmoid(pc._opacity.detach()).max()) <= 0.01 + 1e-6 and float(mom[0].abs().sum()) == 0.0   # [REF :526-530]
