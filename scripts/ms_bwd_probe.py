"""where does mal_loss_multiscale_bwd's time go?  its launch timed back to back (HIP events) with subsets of the outputs"""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
from mal_amd import step, trainer, config, _lib as L
from mal_amd.synthetic import make_batch, to_dicts
B, H, W, sclm = 12, 192, 640, 3
dev = "cuda:0"
config.noise_source = "philox"
batch = make_batch(B, H, W, seed=1)
opt = trainer.default_options(height=H, width=W, batch_size=B, sclm=sclm, distil=False)
inputs, mono_outputs, outputs, leaves = to_dicts(batch, lambda a, t, inv: None, device=dev)
for s in range(1, sclm + 1):
    inputs[("color", 0, s)] = torch.nn.functional.avg_pool2d(batch["color0"], 2 ** s).to(dev)
    for name, outs in (("disp_teacher", mono_outputs), ("disp_student", outputs)):
        leaf = torch.nn.functional.avg_pool2d(batch[name], 2 ** s).to(dev).clone().requires_grad_(True)
        leaves["%s_s%d" % (name, s)] = leaf
        outs[("disp", s)] = leaf
for f, s in ((-1, "m1"), (1, "p1")):
    mono_outputs[("axisangle", 0, f)] = leaves["axisangle_" + s]
    mono_outputs[("translation", 0, f)] = leaves["translation_" + s]
losses, _ = step.loss_step_multiscale(opt, inputs, mono_outputs, outputs, want_maps=False)
node = losses["loss"].grad_fn
while not hasattr(node, "args"):
    node = node.next_functions[0][0]
a = node.args
one = torch.ones(1, device=dev)
a.g_total = one.data_ptr()
bufs = {(n, s): torch.empty(B, 1, H >> s, W >> s, device=dev) for n in range(2) for s in range(4)}
pose = [torch.empty(B, 3, device=dev) for _ in range(4)]
lib = L.load()

def run(want, want_pose=True):
    for s in range(4):
        a.g_disp_teacher[s] = bufs[(0, s)].data_ptr() if want(0, s) else None
        a.g_disp_student[s] = bufs[(1, s)].data_ptr() if want(1, s) else None
    ptrs = [p.data_ptr() if want_pose else None for p in pose]
    a.g_axisangle_m1, a.g_translation_m1, a.g_axisangle_p1, a.g_translation_p1 = ptrs
    for _ in range(3):
        lib.mal_loss_multiscale_bwd(C.byref(a))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.mal_loss_multiscale_bwd(C.byref(a))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 20

print("everything        %.1f us" % run(lambda n, s: True))
print("no pose           %.1f us" % run(lambda n, s: True, False))
print("poses only        %.1f us" % run(lambda n, s: False))
for s in range(4):
    print("scale %d both nets  %.1f us" % (s, run(lambda n, t: t == s, False)))
print("teacher all       %.1f us" % run(lambda n, s: n == 0, False))
print("student all       %.1f us" % run(lambda n, s: n == 1, False))
