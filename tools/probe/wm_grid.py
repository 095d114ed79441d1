import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gaussianprediction_amd import _lib
from gaussianprediction_amd.weights_ops import WeightsModel
N = 1_000_000
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
xyz = ((torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor([1.5, 1.5, 0.5])).to(dev)
m = WeightsModel(12, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
def fwd():
    with torch.no_grad():
        return m(xyz)
for cap in [int(a) for a in sys.argv[1:]]:
    _lib.lib().gp_debug_option(4, cap)
    print(cap, round(timeit(fwd), 4), flush=True)
