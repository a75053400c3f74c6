"""Same-box timing of the whole training step of the harness (bench.py --mode train's TrainStep) under MIOpen / layout
settings: cudnn.benchmark (MIOpen's find mode instead of its immediate-mode heuristics; upstream's train.py sets False),
channels_last parameters + activations.  Prints ms per step (median of 3 x 10 steps after warm-up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(tag, benchmark, channels_last):
    torch.backends.cudnn.benchmark = benchmark
    dev = torch.device("cuda:0")
    st = bench.TrainStep(dev, 1234)
    if channels_last:
        st.h.model.to(memory_format=torch.channels_last)
        st.inputs = {k: (v.contiguous(memory_format=torch.channels_last) if torch.is_tensor(v) and v.dim() == 4 else v)
                     for k, v in st.inputs.items()}
    for _ in range(6):
        st()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            st()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10 * 1e3)
    print("%-40s %.2f ms per step  (%s)" % (tag, sorted(ts)[1], ", ".join("%.2f" % t for t in ts)), flush=True)
    print("   breakdown", {k: round(v, 2) for k, v in st.breakdown_ms(3).items()}, flush=True)
    del st
    torch.cuda.empty_cache()


if __name__ == "__main__":
    which = sys.argv[1:] or ["base", "bench", "cl", "bench_cl"]
    cfg = {"base": (False, False), "bench": (True, False), "cl": (False, True), "bench_cl": (True, True)}
    for w in which:
        try:
            run(w, *cfg[w])
        except Exception as ex:
            print(w, "FAILED", type(ex).__name__, str(ex)[:300], flush=True)
