"""Is the loss path bit-reproducible run to run (it has no floating-point atomics)?  Two runs of the same step on the same
inputs in one process, and the harness' training step twice from the same seeds (MIOpen's atomics are then the only source
of differences)."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mal_amd import _lib, harness, step, trainer
from mal_amd.synthetic import make_batch, to_dicts

dev = "cuda:0"
lib = _lib.load()
for pack_rows in (9, 10):
    lib.mal_set_option(b"pack_rows", pack_rows)
    for (B, H, W) in ((2, 64, 128), (12, 192, 640), (3, 37, 50)):
        batch = make_batch(B, H, W, seed=11)
        g = torch.Generator().manual_seed(5)
        noise = torch.randn(B, 1, H, W, generator=g).to(dev)
        res = []
        for rep in range(3):
            inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=dev)
            for f, s in ((-1, "m1"), (1, "p1")):
                mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
                mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
            opt = trainer.default_options(height=H, width=W, batch_size=B)
            # scribble over freed memory between runs so that a read of uninitialised memory shows
            junk = torch.full((64 << 20,), float("nan") if rep == 1 else 1e30, device=dev); del junk
            losses, _, _ = step.loss_step(opt, inputs, mono_outputs, outputs, noise=noise.clone(), want_maps=False)
            losses["loss"].backward()
            torch.cuda.synchronize()
            res.append((float(losses["loss"].detach()), {k: t.grad.clone() for k, t in leaves.items()}))
        same = all(res[0][0] == r[0] and all(torch.equal(res[0][1][k], r[1][k]) for k in r[1]) for r in res[1:])
        print("pack_rows", pack_rows, (B, H, W), "loss", res[0][0], "bit-identical over 3 runs:", same, flush=True)
        if not same:
            for k in res[0][1]:
                d = max(float((res[0][1][k] - r[1][k]).abs().max()) for r in res[1:])
                print("    ", k, "max diff", d, "scale", float(res[0][1][k].abs().max()))
lib.mal_set_option(b"pack_rows", 10)
outs = []
for rep in range(2):
    torch.manual_seed(0); random.seed(0)
    opt = harness.default_options(batch_size=2, height=64, width=128, no_matching_augmentation=True)
    h = harness.TrainHarness(opt, torch.device(dev), exchange_segments=1)
    inputs = harness.synthetic_inputs(opt, torch.device(dev), seed=11)
    h.model.train()
    random.seed(5); torch.manual_seed(5)
    h.train_step(inputs)
    torch.cuda.synchronize()
    outs.append(h.bucket.flat.detach().clone())
d = outs[0] - outs[1]
print("harness train_step twice: max diff / max", float(d.abs().max()) / float(outs[0].abs().max()), "l2 rel", float(d.double().norm() / outs[0].double().norm()))
