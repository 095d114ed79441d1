"""torch.optim.Adam counts steps PER PARAMETER and passes over a parameter whose .grad is None.  The reference's loop relies on
that without saying so: densify / prune / reset_opacity run between backward and optimizer.step() [REF train.py:164-197] and
replace the per-Gaussian Parameters [REF scene/gaussian_model.py:526-630], so on those iterations Adam skips them -- no update
from that iteration's gradient, and their `step` (the bias corrections) stays one behind the MLP's per event.
FusedAdam keeps one counter and a per-group lag (`hold=`, `lag`, `item_steps`); these tests drive it on CPU tensors beside a
torch.optim.Adam that is fed grad=None on the same steps (the arithmetic of the CPU seam is tests/host_checkers.adam_host_step;
the HIP kernel's per-tensor steps are checked in tests/test_gpu_training_api.py)."""
from types import SimpleNamespace

import pytest
import torch

from gaussianprediction_amd import densify
from gaussianprediction_amd.dist import FlatGradBucket
from gaussianprediction_amd.loss_ops import FusedAdam
import host_checkers

host_checkers.install()

NAMES = ["xyz", "opacity", "df_mlp"]
SHAPES = {"xyz": [(7, 3)], "opacity": [(7, 1)], "df_mlp": [(4, 5), (5,)]}
LRS = {"xyz": 1.6e-4, "opacity": 0.05, "df_mlp": 1e-3}


def _pair(seed=0):
    g = torch.Generator().manual_seed(seed)
    mk = lambda: [{"params": [torch.nn.Parameter(torch.randn(s, generator=g)) for s in SHAPES[n]], "lr": LRS[n], "name": n} for n in NAMES]
    a = mk()
    b = [{"params": [torch.nn.Parameter(p.detach().clone()) for p in grp["params"]], "lr": grp["lr"], "name": grp["name"]} for grp in a]
    ref = torch.optim.Adam(a, lr=0.0, eps=1e-15)
    bucket = FlatGradBucket([p for grp in b for p in grp["params"]])
    return ref, FusedAdam(b, bucket, eps=1e-15), bucket


def _drive(ref, fused, held_by_step, seed=1):
    g = torch.Generator().manual_seed(seed)
    for held in held_by_step:
        for ga, gb in zip(ref.param_groups, fused.param_groups):
            for pa, pb in zip(ga["params"], gb["params"]):
                grad = torch.randn(pa.shape, generator=g)
                pb.grad.copy_(grad)                                   # (views of the flat bucket)
                pa.grad = None if ga["name"] in held else grad        # what a tensor replaced before optimizer.step() looks like
        ref.step()
        fused.step(zero_grad=True, hold=held)


def _assert_same(ref, fused, tol=0.0):
    for ga, gb in zip(ref.param_groups, fused.param_groups):
        for pa, pb in zip(ga["params"], gb["params"]):
            assert torch.allclose(pa.detach(), pb.detach(), rtol=tol, atol=tol * 1e-3), ga["name"]
            st = ref.state.get(pa)
            if st:
                fs = fused.state[pb]
                assert float(fs["step"]) == float(st["step"]), ga["name"]
                assert torch.allclose(fs["exp_avg"], st["exp_avg"], rtol=tol, atol=1e-7) and torch.allclose(fs["exp_avg_sq"], st["exp_avg_sq"], rtol=tol, atol=1e-7)


def test_held_groups_skip_like_grad_none_and_fall_behind():
    ref, fused, bucket = _pair()
    sched = [(), (), ("xyz", "opacity"), (), ("opacity",), (), ()]
    _drive(ref, fused, sched)
    assert fused.step_count == 7 and fused.lag == {"xyz": 1, "opacity": 2}
    _assert_same(ref, fused, tol=2e-6)
    assert float(bucket.flat.abs().max()) == 0.0                       # held gradients were dropped too
    # the state dict carries torch's per-parameter counts, and torch takes it
    sd = fused.state_dict()
    assert [float(sd["state"][k]["step"]) for k in sorted(sd["state"])] == [6.0, 5.0, 7.0, 7.0]
    assert {k: float(v["step"]) for k, v in sd["state"].items()} == {k: float(v["step"]) for k, v in ref.state_dict()["state"].items()}


def test_loaded_counts_are_kept_per_group_and_training_continues_identically():
    ref, fused, _ = _pair(seed=4)
    _drive(ref, fused, [(), ("xyz",), (), ("xyz", "opacity"), ()], seed=5)
    ref2, fused2, _ = _pair(seed=4)
    for (ga, gb, gc, gd) in zip(ref.param_groups, fused.param_groups, ref2.param_groups, fused2.param_groups):
        for pa, pb, pc, pd in zip(ga["params"], gb["params"], gc["params"], gd["params"]):
            pc.data.copy_(pa.data), pd.data.copy_(pa.data)
    ref2.load_state_dict(ref.state_dict())
    fused2.load_state_dict(ref.state_dict())                           # a REFERENCE optimizer's tuple entry, counts differing
    assert fused2.step_count == 5 and fused2.lag == {"xyz": 2, "opacity": 1}
    _drive(ref2, fused2, [(), ("opacity",), ()], seed=6)
    assert fused2.lag == {"xyz": 2, "opacity": 2}
    _assert_same(ref2, fused2, tol=2e-6)


def test_held_groups_follow_the_reference_loops_conditions():
    pg = {"xyz": 0, "f_dc": 0, "f_rest": 0, "opacity": 0, "scaling": 0, "rotation": 0, "motion_feature": 0}
    model = SimpleNamespace(_per_gaussian=lambda: pg,
                            optimizer=SimpleNamespace(param_groups=[{"name": n} for n in list(pg) + ["df_mlp"]]))
    opt = SimpleNamespace(densify_until_iter=15000, densify_from_iter=500, densification_interval=100, opacity_reset_interval=3000)
    every = tuple(sorted(pg))
    assert densify.held_groups(model, 499, opt) == () and densify.held_groups(model, 500, opt) == ()      # "> densify_from_iter"
    assert densify.held_groups(model, 600, opt) == every and densify.held_groups(model, 601, opt) == ()
    assert densify.held_groups(model, 3000, opt) == every                                                    # densify + reset
    assert densify.held_groups(model, 15000, opt) == ()                                                      # "< densify_until_iter"
    assert densify.held_groups(model, 500, opt, white_background=True) == ("opacity",)
    opt.densification_interval = 7
    assert densify.held_groups(model, 3000, opt) == ("opacity",)                                             # a reset without a prune
    model.optimizer.param_groups = [{"name": "s_xyz"}, {"name": "df_mlp"}]                                   # stage 2: nothing to hold
    assert densify.held_groups(model, 600, SimpleNamespace(densify_until_iter=15000, densify_from_iter=500, densification_interval=100,
                                                           opacity_reset_interval=3000)) == ()


def test_fresh_gradient_buffers_take_exactly_one_direct_write():
    """FusedAdam.step(fresh_grad=...) marks a just-zeroed .grad buffer fresh; the FIRST producer of the next backward may write into it
    (deform_ops._input_sink) and every later contribution goes through autograd's accumulate -- which also ends the freshness."""
    from gaussianprediction_amd import grad_sink
    from gaussianprediction_amd.deform_ops import _input_sink
    ref, fused, bucket = _pair(seed=8)
    xyz, opa = fused.param_groups[0]["params"][0], fused.param_groups[1]["params"][0]
    for p_ in (xyz, opa):
        p_.grad.copy_(torch.ones_like(p_))
    fused.step(zero_grad=True, fresh_grad=(xyz, opa))
    assert float(bucket.flat.abs().max()) == 0.0 and getattr(xyz, "_gp_fresh_hook", False)
    # a producer takes the buffer once, the second asker is sent through autograd
    buf = _input_sink(xyz, tuple(xyz.shape))
    assert buf is not None and buf.data_ptr() == xyz.grad.data_ptr()
    assert _input_sink(xyz, tuple(xyz.shape)) is None
    # a shape that is not the leaf's, or a non-leaf, never gets one (and does not consume the mark)
    assert _input_sink(opa, (3, 3)) is None and _input_sink(opa * 2.0, tuple(opa.shape)) is None
    # a contribution through autograd ends the freshness of a buffer nobody took
    (opa * 3.0).sum().backward()
    assert float(opa.grad.min()) == 3.0 and _input_sink(opa, tuple(opa.shape)) is None
    # the next optimizer step marks again (same hook, registered once)
    fused.step(zero_grad=True, fresh_grad=(xyz, opa))
    assert _input_sink(opa, tuple(opa.shape)) is not None
    grad_sink.forget_all()
    assert _input_sink(xyz, tuple(xyz.shape)) is None
