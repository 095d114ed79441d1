"""Train steps keep working across a densify / prune with optimizer-state surgery (fused Adam + flat gradient bucket)."""
import math
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

import gaussianprediction_amd as gpa  # noqa: E402
from gaussianprediction_amd import densify as dn  # noqa: E402
from gaussianprediction_amd.cameras import orbit_cameras  # noqa: E402
from gaussianprediction_amd.renderer import render  # noqa: E402
from gaussianprediction_amd.scene_synth import SceneSpec, make_gaussians, make_keypoints  # noqa: E402
from gaussianprediction_amd.train_step import TrainStep  # noqa: E402


@pytest.mark.parametrize("iteration", [5000, 50000])
def test_steps_densify_steps(iteration):
    dev = "cuda"
    margs = SimpleNamespace(beta=0.1, d=4, w=256, feature_dim=32, second_stage_iteration=30000, third_stage_iteration=40000,
                            jointly_iteration=1000, nearest_num=6, norm_rotation=True, step_opacity=False,
                            step_opacity_iteration=5000, opacity_type="implicit", xyz_noise_iteration=0)
    raw = make_gaussians(SceneSpec(n_gaussians=6000, extent=(1.3, 1.3, 1.3), scale_lo=0.01, scale_hi=0.08, seed=11), device=dev)
    kp, kpf, _, _ = make_keypoints(raw["xyz"], raw["motion_feature"], 48, 6)
    pc = gpa.GaussianModel(3, margs)
    pc.set_inputDim(12, 60)
    pc.create_from_tensors(raw["xyz"], raw["features_dc"], raw["features_rest"], raw["scaling"], raw["rotation"], raw["opacity"],
                           raw["motion_feature"], kp, kpf)
    cams = orbit_cameras(4, 4.0, 0.69, 160, 128, device=dev)
    pipe = SimpleNamespace(convert_SHs_python=False, compute_cov3D_python=False, debug=False)

    def knn_and_weights():
        k2, f2, idx, rw = make_keypoints(pc._xyz.detach(), pc.motion_feature.detach(), 48, 6)
        pc.set_keypoint_weights(rw, idx)

    knn_and_weights()
    with torch.no_grad():
        gts = [render(c, pc, pipe, torch.zeros(3, device=dev), time=torch.tensor([0.3], device=dev), it=iteration)["render"] * 0.9
               for c in cams]
    ts = TrainStep(pc, cams, gts, iteration, schedule=True)
    losses = []
    for i in range(3):
        loss, pkg = ts.step(i)
        dn.track_view(pc, pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"])
        losses.append(float(loss))
    n0 = pc._xyz.shape[0]
    step0 = pc.optimizer.step_count
    if iteration <= 30000:          # the reference densifies in stage 1 [REF train.py:164-177]: densify, then prune
        m_old = pc.adam_moments()[id(pc._xyz)][0].clone()
        n_clone, n_src = pc.densify(1e-7, 0.005, 4.0, 20)
        assert n_clone + n_src > 0 and pc._xyz.shape[0] == n0 + n_clone + n_src
        assert float(pc.max_radii2D.abs().sum()) == 0.0            # reset for every row [REF scene/gaussian_model.py:661]
        m_new = pc.adam_moments()[id(pc._xyz)][0]
        assert float(m_new[n0 - n_src:].abs().sum()) == 0.0 and float(m_new[:n0 - n_src].abs().sum()) > 0 and m_old.shape[0] == n0
        n_pruned = pc.prune(1e-7, 0.005, 4.0, 20)
        assert pc._xyz.shape[0] == n0 + n_clone + n_src - n_pruned
    else:
        pc.reset_opacity()
    assert pc.optimizer.step_count == step0
    if pc._xyz.shape[0] != n0:
        knn_and_weights()
    for i in range(3):
        loss, pkg = ts.step(i)
        losses.append(float(loss))
        assert pkg["radii"].shape[0] == pc._xyz.shape[0]
    assert all(math.isfinite(v) for v in losses)
    assert torch.isfinite(pc._xyz).all() and torch.isfinite(pc._features_rest).all()
