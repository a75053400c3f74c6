"""Temporal-hint producer (N2): HIP kernels vs the CPU checker, one sample of 192x640 with `num` instances,
forward + backward.  Algorithmic bytes per pixel: 2*num mask bytes + 2*12 image + 2*12 out (fwd);
2*num + 1 flag + 2*12 cotangent + 2*12 gradient (bwd)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mal_amd import dyn_utils
from oracle import dyn_oracle as D
from oracle.gen_golden_dyn import make_masks
num, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 12, 192, 640
ml, mn = make_masks(num, H, W, seed=9)
il, inx = torch.rand(3, H, W), torch.rand(3, H, W)
dev = torch.device("cuda:0")
dml, dmn = ml.to(dev), mn.to(dev)
dl, dn = il.to(dev).requires_grad_(True), inx.to(dev).requires_grad_(True)
def gpu():
    a, b = dyn_utils.generate_dynamic_instance(None, None, dml, dmn, dl, dn, False)
    torch.autograd.grad(a.sum() + b.sum(), [dl, dn])
for _ in range(5): gpu()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): gpu()
torch.cuda.synchronize(); tg = (time.perf_counter() - t) / 50
s = torch.cuda.Stream(); g = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    gpu()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    gpu()
for _ in range(5): g.replay()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(200): g.replay()
torch.cuda.synchronize(); tgg = (time.perf_counter() - t) / 200
cl, cn = il.clone().requires_grad_(True), inx.clone().requires_grad_(True)
def cpu():
    a, b = D.generate_dynamic_instance(ml, mn, cl, cn, False)
    torch.autograd.grad(a.sum() + b.sum(), [cl, cn])
cpu(); t = time.perf_counter()
for _ in range(5): cpu()
tc = (time.perf_counter() - t) / 5
px = H * W
bytes_ = px * ((2 * num + 48) + (2 * num + 1 + 48))
print("num=%d  HIP eager %.1f us  HIP graph %.1f us (%.0f GB/s algorithmic)  CPU checker %.1f ms (%d threads)" %
      (num, tg * 1e6, tgg * 1e6, bytes_ / tgg / 1e9, tc * 1e3, torch.get_num_threads()))
