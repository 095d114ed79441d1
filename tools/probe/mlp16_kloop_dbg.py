#!/usr/bin/env python
"""Where the carried stores of the hand-scheduled k-loop differ from the burst form (debugging aid): small row count, element map."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gaussianprediction_amd as gpa
from gaussianprediction_amd import _lib
from gaussianprediction_amd.deform_ops import FusedMlp16
F = 6
torch.manual_seed(0)
net = gpa.Deformable_Field(32 + 60 + 2 * F, output_dim=7, d=4, w=256, precision="fp32s").cuda()
L = _lib.lib()
class Ctx:
    def save_for_backward(self, *a): self.saved = a
def run_raw(n, bits):
    feat = (torch.rand(n, 32, device="cuda", generator=torch.Generator("cuda").manual_seed(1)) - 0.5).requires_grad_(True)
    xyz = (torch.rand(n, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) * 2.6 - 1.3)
    t = torch.tensor([0.3], device="cuda")
    _lib.check(L.gp_debug_option(9, bits), "opt")
    ctx = Ctx()
    out = FusedMlp16.forward(ctx, feat, xyz, t, 10, F, "fp32s", None, *net._wb())
    torch.cuda.synchronize()
    return out, ctx.saved[3], ctx.saved[4], ctx.saved[5]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = run_raw(n, 64)[2].view(torch.int16).view(4, -1, 512, 16)      # [layer][row block][feature][row]
b = run_raw(n, 0)[2].view(torch.int16).view(4, -1, 512, 16)
d = (a != b)
print("differing per layer:", d.sum(dim=(1, 2, 3)).tolist(), "of", a[0].numel())
for l in range(4):
    if d[l].any():
        print("layer", l, "per row-block%4:", [int(d[l][k::4].sum()) for k in range(4)])
        print("   per feature group of 32:", d[l].sum(dim=(0, 2)).view(16, 32).sum(1).tolist())
        print("   per row in block:", d[l].sum(dim=(0, 1)).tolist())
        idx = d[l].nonzero()[:8]
        for rb, f, r in idx.tolist():
            # where does the value b has come from in a?
            v = b[l, rb, f, r]
            src = (a[l, rb // 4 * 4: rb // 4 * 4 + 4] == v).nonzero()[:4].tolist()
            print(f"   (rb {rb}, f {f}, r {r}): burst {int(a[l, rb, f, r])} carried {int(v)}  found in burst at (rb%4, f, r) {src}")
        break
