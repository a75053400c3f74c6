#!/usr/bin/env python
"""Where the host time of an EAGER headline step goes (bench.py's eager_ms_per_step: the real temporal-hint producer cannot be
captured into a graph): cProfile over 200 eager steps, top entries by cumulative and by own time.
    python scripts/eager_profile.py [--mode step|distil]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "step"
    dev = torch.device("cuda", 0)
    step = bench.Step(dev, 1234, mode)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    print("eager ms/step: %.4f" % (1e3 * (time.perf_counter() - t0) / 200))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    pr.disable()
    for key in ("cumulative", "tottime"):
        st = pstats.Stats(pr)
        st.sort_stats(key).print_stats(28)


if __name__ == "__main__":
    main()
