#!/usr/bin/env python
"""A/B of the hand-scheduled k-loops of the large-row 16-bit MLP (tools/gen_mlp16_kloop.py) against the compiler's loops, on one box,
alternating: gp_debug_option(9, 64) selects the compiler's loops + burst stores in the SAME binary.  Checks first that both forms give
bit-identical outputs, saved tensors and ReLU masks (same MFMA order per accumulator), then times forward and backward.
    python tools/probe/mlp16_kloop_ab.py [precision] [rows] [extra ablate bits]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib
from gaussianprediction_amd.deform_ops import FusedMlp16

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32s"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
F = 6
torch.manual_seed(0)
net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256, precision=prec).cuda()
L = _lib.lib()


class Ctx:
    def save_for_backward(self, *a):
        self.saved = a


def run_raw(n, bits):
    feat = (torch.rand(n, 32, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) - 0.5).requires_grad_(True)
    xyz = (torch.rand(n, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) * 2.6 - 1.3)
    t = torch.tensor([0.3], device="cuda")
    wb = net._wb()
    _lib.check(L.gp_debug_option(9, bits), "opt")
    ctx = Ctx()
    out = FusedMlp16.forward(ctx, feat, xyz, t, 10, F, prec, None, *wb)
    torch.cuda.synchronize()
    return out, ctx.saved[3], ctx.saved[4], ctx.saved[5]


for n in (rows, 64 * 3 + 37, 4096 + 5):
    a = run_raw(n, 64)
    b = run_raw(n, 0)
    same = [bool(torch.equal(x.view(torch.int16) if x.dtype in (torch.float16, torch.bfloat16) else x, y.view(torch.int16) if y.dtype in (torch.float16, torch.bfloat16) else y)) for x, y in zip(a, b)]
    print(f"rows {n}: out / xT / hT / masks bit-identical between the two forms: {same}", flush=True)



def grads(n, bits):
    g = torch.Generator("cuda").manual_seed(5)
    f = (torch.rand(n, 32, device="cuda", generator=g) - 0.5).requires_grad_(True)
    x = (torch.rand(n, 3, device="cuda", generator=g) * 2.6 - 1.3).requires_grad_(True)
    gy = torch.randn(n, 7, device="cuda", generator=g)
    _lib.check(L.gp_debug_option(9, bits), "opt")
    net.zero_grad(set_to_none=True)
    net.forward_fused(f, x, torch.tensor([0.3], device="cuda"), 10, F).backward(gy)
    torch.cuda.synchronize()
    return [f.grad, x.grad] + [q.grad.clone() for q in net.parameters()]


for n in (65536 + 77, 64 * 3 + 37 + 4096):
    a, b = grads(n, 64), grads(n, 0)
    print(f"rows {n}: backward: dfeature / dxyz bit-identical {torch.equal(a[0], b[0])} {torch.equal(a[1], b[1])};  weight-gradient rel. differences (atomic order only) "
          + " ".join(f"{float((u - v).norm() / u.norm()):.1e}" for u, v in zip(a[2:], b[2:])), flush=True)

feat = (torch.rand(rows, 32, device="cuda") - 0.5).requires_grad_(True)
xyz = (torch.rand(rows, 3, device="cuda") * 2.6 - 1.3).requires_grad_(True)
t = torch.tensor([0.3], device="cuda")


def timed(bits, backward=True, reps=6):
    _lib.check(L.gp_debug_option(9, bits), "opt")
    for _ in range(3):
        y = net.forward_fused(feat, xyz, t, 10, F)
        if backward:
            y.sum().backward()
    torch.cuda.synchronize()
    _lib.profile_enable(True); _lib.profile_collect()
    for _ in range(reps):
        y = net.forward_fused(feat, xyz, t, 10, F)
        if backward:
            y.sum().backward()
    torch.cuda.synchronize()
    p = _lib.profile_collect(); _lib.profile_enable(False)
    return {k: v[1] / v[0] * 1e3 for k, v in p.items() if k.startswith("mlp16")}


for rnd in range(3):
    for bits, what in ((64, "compiler loops, burst stores"), (0, "hand-scheduled, carried stores"), (64 + 1, "compiler loops, no saved-tensor stores"),
                       (1, "hand-scheduled, no saved-tensor stores")):
        r = timed(bits | extra, backward=not (bits & 1))
        print(f"{prec} rows {rows} round {rnd} [{what:40s}] " + "  ".join(f"{k} {v:8.1f} us" for k, v in sorted(r.items())), flush=True)
_lib.check(L.gp_debug_option(9, 0), "opt")
