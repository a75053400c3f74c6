"""whole training step of the harness under MIOpen knobs: benchmark (find) mode, channels_last"""
import os, sys, time, torch
sys.path.insert(0, ".")
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "base"
if "bench" in mode:
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
ts = bench.TrainStep(dev, 1234)
if "cl" in mode:
    for m in ts.h.models.values() if hasattr(ts.h, "models") else []:
        m.to(memory_format=torch.channels_last)
for _ in range(6):
    ts()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    ts()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 10
print(mode, "%.2f ms/step -> %.1f images/s" % (dt * 1e3, bench.B / dt), ts.breakdown_ms(3))
