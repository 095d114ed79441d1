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


# ---- a loop in the REFERENCE's order: backward -> densify / reset_opacity / prune -> optimizer.step() [REF train.py:164-197] -------
def test_reference_order_loop_updates_the_survivors_and_passes_over_the_replaced_tensors():
    """The advisor's round-4 finding.  In the reference's order the surgery runs on a model whose gradient of this iteration has
    not been consumed yet.  torch.optim.Adam then (a) updates every parameter that SURVIVES as an object -- the MLP, the keypoints --
    from that gradient and (b) passes over the replaced per-Gaussian tensors (.grad None), whose step count falls one behind.
    Rebuilding bucket + optimizer must not throw (a) away, and must not turn (b) into an update with a zero gradient (moments
    decaying, parameters coasting on momentum)."""
    from test_densify import _fake_adam_state, _setup
    pc = _setup(n=40)
    _fake_adam_state(pc, step=17)
    mlp = list(pc.df_model.parameters())
    # the twin: torch.optim.Adam over clones of the MLP, same state
    twin_p = [torch.nn.Parameter(p.detach().clone()) for p in mlp]
    twin = torch.optim.Adam([{"params": twin_p, "lr": 0.0, "name": "df_mlp"}], lr=0.0, eps=1e-15)
    mom = pc.adam_moments()
    for q, p in zip(twin_p, mlp):
        twin.state[q] = {"step": torch.tensor(17.0), "exp_avg": mom[id(p)][0].clone(), "exp_avg_sq": mom[id(p)][1].clone()}
    lr_mlp = next(g["lr"] for g in pc.optimizer.param_groups if g["name"] == "df_mlp")
    twin.param_groups[0]["lr"] = lr_mlp
    g = torch.Generator().manual_seed(0)
    for p in pc.bucket.params:                             # "loss.backward()": every optimized tensor has a gradient
        p.grad.copy_(torch.randn(p.shape, generator=g) * 1e-2)
    for q, p in zip(twin_p, mlp):
        q.grad = p.grad.detach().clone()
    # ---- the reference's calls between backward and step
    n = pc._xyz.shape[0]
    pc.denom += 1
    pc.xyz_gradient_accum[[1, 4, 7]] = 1.0
    with torch.no_grad():
        pc._scaling[[1, 4]] = float(torch.log(torch.tensor(0.2)))
        pc._scaling[7] = float(torch.log(torch.tensor(0.01)))
        pc._opacity[[1, 4, 7]] = 2.0
    pc.densify(0.0002, 0.005, 5.0, None, generator=torch.Generator().manual_seed(1))
    pc.reset_opacity()
    pc.prune(0.0002, 0.005, 5.0, None)
    per_g = set(pc._per_gaussian().keys())
    assert pc.optimizer.pending_hold == per_g & {gr["name"] for gr in pc.optimizer.param_groups}
    before = {k: v.detach().clone() for k, v in pc._per_gaussian().items()}
    mom_before = {k: tuple(t.clone() for t in pc.adam_moments()[id(v)]) for k, v in pc._per_gaussian().items()}
    pc.optimizer.step()                                    # train.py:196
    pc.optimizer.zero_grad(set_to_none=True)               # train.py:197
    twin.step()
    # (a) the MLP took this iteration's gradient, exactly as torch.optim.Adam does
    for q, p in zip(twin_p, mlp):
        torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-6, atol=1e-8)      # (one ulp of the parameter)
    assert float(pc.optimizer.state[mlp[0]]["step"]) == 18.0
    # (b) the replaced tensors: values and moments untouched, count one behind, nothing pending any more
    for k, v in pc._per_gaussian().items():
        assert torch.equal(v.detach(), before[k]), k
        m, s = pc.adam_moments()[id(v)]
        assert torch.equal(m, mom_before[k][0]) and torch.equal(s, mom_before[k][1]), k
        assert float(pc.optimizer.state[v]["step"]) == 17.0, k
    assert pc.optimizer.pending_hold == set() and float(pc.bucket.flat.abs().max()) == 0.0
    assert pc.optimizer.lag == {k: 1 for k in per_g if k in {gr["name"] for gr in pc.optimizer.param_groups}}
    # the next iteration is an ordinary one: everybody steps, the per-Gaussian groups with THEIR count (18, the MLP 19)
    for p in pc.bucket.params:
        p.grad.copy_(torch.randn(p.shape, generator=g) * 1e-2)
    pc.optimizer.step()
    assert float(pc.optimizer.state[pc._xyz]["step"]) == 18.0 and float(pc.optimizer.state[mlp[0]]["step"]) == 19.0
    assert not torch.equal(pc._xyz.detach(), before["xyz"])


def test_this_packages_own_order_holds_nothing_back():
    """update first, operate afterwards (TrainStep + densify.py): no unconsumed gradient at surgery time -> nothing pending."""
    from test_densify import _fake_adam_state, _setup
    pc = _setup(n=40)
    _fake_adam_state(pc, step=3)
    g = torch.Generator().manual_seed(0)
    for p in pc.bucket.params:
        p.grad.copy_(torch.randn(p.shape, generator=g) * 1e-2)
    pc.optimizer.step()                                    # consumes and zeroes
    pc.denom += 1
    pc.xyz_gradient_accum[:5] = 1.0
    pc.densify(0.0002, 0.005, 5.0, None, generator=torch.Generator().manual_seed(1))
    pc.reset_opacity()
    pc.prune(0.0002, 0.005, 5.0, None)
    assert pc.optimizer.pending_hold == set() and pc.optimizer.lag == {}


def test_group_names_are_unique_keys():
    """`lag` / `hold` are keyed by group name (the advisor's round-4 note): unnamed groups are given one, duplicates are refused."""
    p = [torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(4))]
    groups = [{"params": [p[0]], "lr": 0.1}, {"params": [p[1]], "lr": 0.2}]
    opt = FusedAdam(groups, FlatGradBucket(p))
    assert [g["name"] for g in opt.param_groups] == ["group0", "group1"]
    with pytest.raises(ValueError):
        FusedAdam([{"params": [p[0]], "lr": 0.1, "name": "a"}, {"params": [p[1]], "lr": 0.2, "name": "a"}], FlatGradBucket(p))
