"""-m gpu: the stage-3 train step through ONE call of the library (gp_train_step_run; fused_step.FusedStage3) against the same step
through the autograd graph (render() -> L1SSIMLoss -> backward -> FusedAdam.step).  The C entry calls the library's own entry points
in the order the graph runs them, so everything must agree up to the summation order of the backward's atomics: gradients (read
off the first Adam moment at zero learning rate), losses, and parameters after real updates.  [REF train.py:101-133, 196-197]"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gaussianprediction_amd.train_step import TrainStep  # noqa: E402
from test_gpu_training_api import build  # noqa: E402

ZERO = dict(xyz=0.0, f_dc=0.0, opacity=0.0, scaling=0.0, rotation=0.0, kpts=0.0, mlp=0.0)


def _run(fused, lrs, steps, time_offset=False, n=6000, early_adam=True):
    pc, cams, gts, raw, rw, idx, args = build(n=n)
    ts = TrainStep(pc, cams, gts, 50000, lrs=lrs, speculative=True, fused=fused)
    ts.early_adam = early_adam
    pre = len(cams) + TrainStep.SPEC_SLOTS                     # exact-mode set-up steps (graph path in both runs)
    losses = []
    for i in range(pre + steps):
        off = torch.full((1,), 0.01 * (i % 3), device="cuda") if time_offset else None
        loss, pkg = ts.step(i, time_offset=off)
        losses.append(loss)
    torch.cuda.synchronize()
    sd = pc.optimizer.state_dict()
    names = [n_ for n_, p in pc.named_parameters() if p.requires_grad]
    return dict(ts=ts, pc=pc, loss=[float(x) for x in losses], pkg={k: v.clone() if torch.is_tensor(v) else v for k, v in pkg.items()},
                vs_grad=pkg["viewspace_points"].grad.clone(), params={n_: p.detach().clone() for n_, p in pc.named_parameters()},
                state=sd["state"], names=names, steps=pre + steps, redone=ts.redone)


def test_fused_step_equals_the_graph_step_gradients():
    """Zero learning rates: the parameters stand still, every step sees the same model, and the Adam moments after k steps are a
    fixed function of the k gradients -- so equal moments = equal gradients, for every optimized tensor, MLP weights included."""
    a = _run(True, ZERO, 6)
    b = _run(False, ZERO, 6)
    assert a["ts"].fused_steps == 6 and b["ts"].fused_steps == 0 and a["redone"] == b["redone"] == 0
    assert np.allclose(a["loss"], b["loss"], rtol=2e-6, atol=1e-7), (a["loss"], b["loss"])
    for k in a["params"]:
        assert torch.equal(a["params"][k], b["params"][k]), k           # lr = 0: nothing moved in either
    assert a["state"].keys() == b["state"].keys()
    for k in a["state"]:
        for kk in ("exp_avg", "exp_avg_sq"):
            x, y = a["state"][k][kk], b["state"][k][kk]
            e = float((x - y).norm() / y.norm().clamp_min(1e-30))
            assert e < 2e-5, (k, kk, e)
        assert float(a["state"][k]["step"]) == float(b["state"][k]["step"])
    # the result dict has render()'s shape and contents
    for k in ("render", "radii", "visibility_filter", "depth", "tidx"):
        x, y = a["pkg"][k], b["pkg"][k]
        assert x.shape == y.shape and x.dtype == y.dtype, k
        assert torch.equal(x, y) if x.dtype != torch.float32 else torch.allclose(x, y, atol=1e-6), k
    e = float((a["vs_grad"] - b["vs_grad"]).norm() / b["vs_grad"].norm())
    assert e < 2e-5, e


def test_fused_step_trains_like_the_graph_step():
    """The reference's learning rates: after 10 updates the two paths' parameters differ by what two runs of ONE path differ by
    (atomics in the backward + Adam's eps = 1e-15): the typical element a few percent of one learning-rate step."""
    a = _run(True, None, 10, time_offset=True)
    b = _run(False, None, 10, time_offset=True)
    assert a["ts"].fused_steps == 10
    assert np.abs(np.array(a["loss"]) - np.array(b["loss"])).max() < 2e-4
    lr = {}
    for g in b["pc"].optimizer.param_groups:
        for p in g["params"]:
            lr[id(p)] = max(float(g["lr"]), 1e-7)
    for (k, x), (_, p) in zip(a["params"].items(), b["pc"].named_parameters()):
        if id(p) not in lr:
            continue
        d = (x - b["params"][k]).abs().flatten()
        assert float(d.median()) <= 0.25 * lr[id(p)] + 1e-8 and float(d.max()) <= 2 * 10 * lr[id(p)] + 1e-6, (k, float(d.median()), float(d.max()), lr[id(p)])
    assert not torch.equal(a["params"]["_xyz"], build(n=6000)[0]._xyz.detach())        # (it did move)


def test_fused_step_falls_back_when_the_step_has_another_shape():
    """Held groups, several views per step, exact-mode binning: the graph path, silently, with the same results as ever -- and back
    to the fused path afterwards, with the gradient buffers in the state it expects."""
    pc, cams, gts, raw, rw, idx, args = build(n=3000)
    ts = TrainStep(pc, cams, gts, 50000, speculative=True, fused=True)
    pre = len(cams) + TrainStep.SPEC_SLOTS
    for i in range(pre):
        ts.step(i)
    assert ts.fused_steps == 0
    ts.step(pre)
    assert ts.fused_steps == 1
    ts.step(pre + 1, hold=("xyz",))                       # a held group: graph path
    assert ts.fused_steps == 1 and pc.optimizer.lag == {"xyz": 1}
    loss, pkg = ts.step(pre + 2)                          # lagging step counts: still fused (per-tensor steps)
    assert ts.fused_steps == 2 and bool(torch.isfinite(loss))
    sd = pc.optimizer.state_dict()
    steps = {g["name"]: float(sd["state"][g["params"][0]]["step"]) for g in sd["param_groups"]}
    assert steps["xyz"] == steps["f_dc"] - 1
    pc.set_keypoint_weights(rw.clone(), idx)              # new weights tensor: the plan is rebuilt, not reused with a dead pointer
    ts.step(pre + 3)
    assert ts.fused_steps == 3
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p).all()) for p in pc.parameters())


# ---- view-parallel: the same call with the exchange issued at its hook points ---------------------------------------------------
def _rank_fused(rank, world, port, out_dir, fused, steps):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pc, cams, gts, raw, rw, idx, args = build(n=5000)
    pc.bucket_small_numel = 12000            # the SH pair and the geometry get regions of their own, as at 1 M Gaussians
    ts = TrainStep(pc, cams, gts, 50000, speculative=True, fused=fused)
    assert ts.sharded and type(ts.reducer).__name__ == "ShardedExchange"
    pre = len(cams) + TrainStep.SPEC_SLOTS
    losses = []
    for i in range(pre + steps):
        loss, pkg = ts.step(i * world + rank)
        losses.append(float(loss))
    ts.sync_params()
    torch.cuda.synchronize()
    sd = pc.optimizer.state_dict()
    lr_of = {id(p): float(g["lr"]) for g in pc.optimizer.param_groups for p in g["params"]}
    torch.save({"params": {n: p.detach().cpu() for n, p in pc.named_parameters()}, "loss": losses, "fused_steps": ts.fused_steps,
                "lr": {n: max(lr_of.get(id(p), 0.0), 1e-7) for n, p in pc.named_parameters()},
                "state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                "radii": pkg["radii"].cpu(), "vis": pkg["visibility_filter"].cpu()}, os.path.join(out_dir, f"f{int(fused)}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fused_step_equals_the_two_rank_graph_step(tmp_path):
    """Two ranks (one GPU, gloo), sharded exchange: reduce-scatter per region at the library call's hook points -> Adam on this rank's
    slices inside the call -> asynchronous all-gather; the SH region chained to the side stream.  Rank-identical parameters, and the
    trajectory of the graph path."""
    import os
    import socket
    import torch.multiprocessing as mp
    steps = 6
    for fused in (True, False):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(_rank_fused, args=(2, port, str(tmp_path), fused, steps), nprocs=2, join=True)
    r = {(f, k): torch.load(os.path.join(tmp_path, f"f{f}_{k}.pt"), weights_only=False) for f in (0, 1) for k in (0, 1)}
    assert r[(1, 0)]["fused_steps"] == steps and r[(1, 1)]["fused_steps"] == steps and r[(0, 0)]["fused_steps"] == 0
    for f in (0, 1):
        for k in r[(f, 0)]["params"]:
            assert torch.equal(r[(f, 0)]["params"][k], r[(f, 1)]["params"][k]), (f, k)        # both ranks hold the same model
        assert torch.equal(r[(f, 0)]["radii"], r[(f, 1)]["radii"]) and torch.equal(r[(f, 0)]["vis"], r[(f, 1)]["vis"])
    a, b = r[(1, 0)], r[(0, 0)]
    assert np.abs(np.array(a["loss"]) - np.array(b["loss"])).max() < 2e-4
    assert {k: float(v["step"]) for k, v in a["state"].items()} == {k: float(v["step"]) for k, v in b["state"].items()}
    for k in a["state"]:
        x, y = a["state"][k]["exp_avg"], b["state"][k]["exp_avg"]
        assert float((x - y).norm() / y.norm().clamp_min(1e-30)) < 2e-2, k            # (six real updates apart: not bit-equal, the same walk)
    for k, x in a["params"].items():          # in units of the tensor's learning rate (run-to-run noise: profiles/r05_order_noise.txt)
        d, lr = (x - b["params"][k]).abs(), a["lr"][k]
        assert float(d.median()) <= 0.25 * lr + 1e-7 and float(d.max()) <= 2 * steps * lr + 1e-6, (k, float(d.median()), float(d.max()), lr)


def test_adam_update_riding_in_the_mlp_backward_launch_changes_nothing():
    """gp_step_update.adam_early_mask: the per-Gaussian tensors' Adam update rides in the launch of the keypoint MLP's data backward (the
    rider of csrc/loss_adam_kernels.h; round 5: a second stream).  Same updates as the single launch behind the backward: parameters
    after real steps agree as two runs of one path do, the step counts are equal, and the parameters read right after a step on the
    caller's stream are the updated ones."""
    from gaussianprediction_amd import deform_ops
    deform_ops.FORCE_ROW_TILES = True        # (the rider travels with the 16-row kernels only: beside the feature-split ones it measured slower)
    try:
        a = _run(True, None, 8, early_adam=True)
        b = _run(True, None, 8, early_adam=False)
    finally:
        deform_ops.FORCE_ROW_TILES = False
    assert a["ts"]._fused_plan.upd.adam_early_mask != 0 and b["ts"]._fused_plan.upd.adam_early_mask == 0
    assert a["ts"].fused_steps == b["ts"].fused_steps == 8
    assert np.abs(np.array(a["loss"]) - np.array(b["loss"])).max() < 2e-4
    lr = {}
    for g in b["pc"].optimizer.param_groups:
        for p in g["params"]:
            lr[id(p)] = max(float(g["lr"]), 1e-7)
    for (k, x), (_, p) in zip(a["params"].items(), b["pc"].named_parameters()):
        if id(p) in lr:
            d = (x - b["params"][k]).abs().flatten()
            assert float(d.median()) <= 0.25 * lr[id(p)] + 1e-8, (k, float(d.median()), lr[id(p)])
    for k in a["state"]:
        assert float(a["state"][k]["step"]) == float(b["state"][k]["step"])
    # one more step, and the parameters read on the caller's stream WITHOUT a device synchronisation are the updated ones
    ts, pc = a["ts"], a["pc"]
    before = pc._xyz.detach().clone()
    ts.step(a["steps"])
    after = pc._xyz.detach().clone()          # (enqueued on the caller's stream behind the call)
    torch.cuda.synchronize()
    assert torch.equal(after, pc._xyz.detach()) and not torch.equal(before, after)
