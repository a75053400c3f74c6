"""N1: a trainer-compatible harness around the loss path -- ``RepDepth`` (mal_amd.networks),
``process_batch`` in upstream's order (manydepth/trainer.py:555-644, ``--distil`` [``--temporal``]), the adaptive depth-bin
tracker (:75-99), ``LossBalancing`` (:640-642), Adam + StepLR / ``WarmupStepLRScheduler``
(:193-230, lr_scheduler.py:30-70), checkpoints ``model.pth`` / ``track.pth`` / ``adam.pth`` (:1605-1636,
1667-1730) and the data-parallel exchange: the parameter gradients live in ONE flat buffer that is all-reduced over RCCL
in a few large pieces launched from inside the backward as they complete (mal_amd.dp.FlatGradBucket) instead of
accelerate's DDP + per-step barrier.

The loss half of ``process_batch`` is ``mal_amd.step.loss_step`` (6 HIP kernels); the networks run through
torch.nn (MIOpen).  Options follow manydepth/options.py names.
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from . import dp, loss_utils, networks
from .step import loss_step


def default_options(**kw):
    """the fields of manydepth/options.py the harness reads, with upstream's defaults (README flags: --distil)"""
    o = dict(height=192, width=640, batch_size=12, min_depth=0.1, max_depth=100.0, frame_ids=[0, -1, 1], scales=[0],
             sclm=0, num_layers=18, weights_init="scratch", depth_binning="linear", num_depth_bins=96,
             num_matching_frames=1, use_future_frame=False, pose_cnn=False, dc=False, distil=True, no_ens=False,
             temporal=False, main_temporal=False, dual_distil=False, learn_ens=False, no_ssim=False, loss_blc=False,
             disable_automasking=False, no_matching_augmentation=False, notadabins=False, learning_rate=1e-4,
             scheduler_step_size=15, warmup_steps=0, decay_steps=1000, freeze_teacher_epoch=15,
             num_train_data=39810)  # KITTI eigen_zhou train split size (splits/eigen_zhou/train_files.txt)
    o.update(kw)
    return SimpleNamespace(**o)


class DepthBinTracker:
    """trainer.py:75-99 without torchmetrics: exponential averages of the teacher's depth range; rank-local
    (upstream reduces min/max across ranks when ``compute()`` is called, SURVEY.md 8e lets it stay local)."""

    def __init__(self, opt_min_depth):
        self.min_depth, self.max_depth = torch.tensor(0.1), torch.tensor(10.0)
        self.opt_min_depth = opt_min_depth

    def update(self, mono_depth):
        d = mono_depth.detach()
        lo = d.amin(dim=(-1, -2)).mean()
        hi = d.amax(dim=(-1, -2)).mean()
        lo = torch.clamp(lo * 0.9, min=self.opt_min_depth)
        hi = hi * 1.1
        dev = lo.device
        self.max_depth = self.max_depth.to(dev) * 0.99 + hi * 0.01
        self.min_depth = self.min_depth.to(dev) * 0.99 + lo * 0.01

    def compute(self):
        return self.min_depth, self.max_depth

    def load(self, min_depth, max_depth):
        self.min_depth, self.max_depth = torch.as_tensor(min_depth).float(), torch.as_tensor(max_depth).float()


class WarmupStepLRScheduler:
    """lr_scheduler.py:30-70: linear warm-up to ``peak_lr``, then x ``decay_scale`` every ``decay_steps`` updates"""

    def __init__(self, optimizer, init_lr, peak_lr, warmup_steps, decay_steps, decay_scale=0.1):
        self.optimizer, self.init_lr = optimizer, init_lr
        self.warmup_rate = (peak_lr - init_lr) / warmup_steps if warmup_steps else 0
        self.warmup_steps, self.decay_steps, self.decay_scale = warmup_steps, decay_steps, decay_scale
        self.update_steps, self.lr = 1, init_lr

    def _set(self, lr):
        for g in self.optimizer.param_groups:
            g["lr"] = lr
        self.lr = lr

    def step(self):
        if self.update_steps < self.warmup_steps:
            self._set(self.init_lr + self.warmup_rate * self.update_steps)
        if self.update_steps > self.warmup_steps and self.update_steps % self.decay_steps == 0:
            self._set(self.lr * self.decay_scale)
        self.update_steps += 1

    def get_last_lr(self):
        return [self.lr]


class StageTimer:
    """device-side stage marks of one training step (network, loss-path, all-reduce and optimizer time
    reported separately); events on the current stream, read back after a synchronize"""

    def __init__(self):
        self.marks = []

    def mark(self, name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append((name, e))

    def result_ms(self):
        torch.cuda.synchronize()
        return {n: self.marks[i - 1][1].elapsed_time(e) for i, (n, e) in enumerate(self.marks) if i}


class TrainHarness:
    def __init__(self, opt, device, process_group=None, image_synthesis=None, exchange_segments=4):
        """``image_synthesis(inputs, outputs, scale) -> has_ins``: the temporal hint's producer (``--temporal``; upstream
        binds dyn_utils.image_synthesis to the segmenter and the matcher, trainer.py:1161-1165).  ``exchange_segments``:
        pieces of the flat gradient buffer that are all-reduced from inside the backward (1 = one all-reduce after it)."""
        self.opt, self.device = opt, torch.device(device)
        self.image_synthesis = image_synthesis
        if getattr(opt, "temporal", False) and image_synthesis is None:
            raise ValueError("TrainHarness: opt.temporal needs image_synthesis(inputs, outputs, scale) -> has_ins")
        self.model = networks.RepDepth(opt).to(self.device)
        self.params = [p for p in self.model.parameters() if p.requires_grad]
        self.optimizer = torch.optim.Adam(self.params, opt.learning_rate)
        if getattr(opt, "warmup_steps", 0):
            self.scheduler = WarmupStepLRScheduler(self.optimizer, 1e-7, opt.learning_rate, opt.warmup_steps, opt.decay_steps)
        else:
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, opt.scheduler_step_size, 0.1)
        self.tracker = DepthBinTracker(opt.min_depth)
        self.loss_blc = loss_utils.LossBalancing(2, opt.num_train_data, opt.batch_size) if opt.loss_blc else None
        # gradients of all trainable parameters as views of one buffer: one all-reduce per step
        self.bucket = dp.FlatGradBucket(self.params, process_group, segments=exchange_segments)
        self.step_count = 0
        self.issued_inside_backward = 0

    # ---- trainer.py:555-644, --distil
    def process_batch(self, inputs, index_iter=0, timer=None):
        min_depth, max_depth = self.tracker.compute()
        mono_outputs, outputs = self.model(inputs, min_depth, max_depth)
        if timer is not None:
            timer.mark("networks_fwd")
        # with --loss_blc the total is bs * sum_i w_i L_i (loss_utils.py:303-318), formed on the device
        w_list = list(self.loss_blc.w_list) if self.loss_blc is not None else None
        losses, loss_list, maps = loss_step(self.opt, inputs, mono_outputs, outputs, w_list=w_list,
                                            image_synthesis=self.image_synthesis)
        if timer is not None:
            timer.mark("loss_path_fwd")
        if not self.opt.notadabins and not self.model.freeze_tp:
            # ("mono_depth", 0, 0) of generate_images_pred: 1 / (1/max + (1/min - 1/max) * disp)   (layers.py:14-23)
            lo, hi = 1.0 / self.opt.max_depth, 1.0 / self.opt.min_depth
            self.tracker.update(1.0 / (lo + (hi - lo) * mono_outputs[("disp", 0)].detach()))
        if self.loss_blc is not None:  # host-side bookkeeping of the running scores and the re-weighting
            self.loss_blc.compute_loss(loss_list, index_iter)
            losses["w_ori"], losses["w_distil"] = self.loss_blc.update_weight(index_iter, 0.0)
        return outputs, losses

    def train_step(self, inputs, timer=None):
        mark = timer.mark if timer is not None else (lambda name: None)
        self.model.train()
        mark("start")
        self.bucket.begin_step()
        outputs, losses = self.process_batch(inputs, self.step_count, timer)
        losses["loss"].backward()  # N>1: the bucket's pieces are all-reduced from inside it as they complete (dp.py)
        mark("loss_path_bwd+networks_bwd")
        self.issued_inside_backward = self.bucket.finish()
        mark("grad_all_reduce")
        self.optimizer.step()
        mark("adam")
        self.step_count += 1
        return losses

    def exchange_note(self):
        b = self.bucket
        return ("%d parameters (%.0f MB fp32) in one flat buffer, %d piece(s); last step: %d issued from inside the backward, "
                "world size %d" % (b.flat.numel(), b.flat.numel() * 4 / 1e6, len(b.bounds), self.issued_inside_backward, b.world_size))

    def end_epoch(self):
        self.scheduler.step()

    # ---- trainer.py:1605-1636, 1667-1730
    def save(self, folder):
        os.makedirs(folder, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(folder, "model.pth"))
        lo, hi = self.tracker.compute()
        torch.save({"height": self.opt.height, "width": self.opt.width, "min_depth_bin": lo.cpu(), "max_depth_bin": hi.cpu()},
                   os.path.join(folder, "track.pth"))
        torch.save(self.optimizer.state_dict(), os.path.join(folder, "adam.pth"))

    def load(self, folder):
        self.model.load_state_dict(torch.load(os.path.join(folder, "model.pth"), map_location="cpu"), strict=False)
        track = torch.load(os.path.join(folder, "track.pth"), map_location="cpu")
        self.tracker.load(track.get("min_depth_bin"), track.get("max_depth_bin"))
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            self.optimizer.load_state_dict(torch.load(adam, map_location="cpu"))
        assert self.bucket.check_views(), "loading must not replace the flat gradient views"


def synthetic_inputs(opt, device, seed=1234):
    """a KITTI-shaped batch with the keys ``RepDepth.forward`` and the loss path read (mono_dataset.py:150-200)"""
    from .synthetic import make_batch
    b = make_batch(opt.batch_size, opt.height, opt.width, seed=seed)
    mv = lambda t: t.to(device).contiguous()
    inputs = {}
    for f, k in ((0, "color0"), (-1, "color_m1"), (1, "color_p1")):
        inputs[("color", f, 0)] = mv(b[k])
        inputs[("color_aug", f, 0)] = mv(b[k])
    inputs[("K", 0)], inputs[("inv_K", 0)] = mv(b["K"]), mv(b["inv_K"])
    K2 = b["K"].clone()
    K2[:, 0] /= 4
    K2[:, 1] /= 4
    inputs[("K", 2)], inputs[("inv_K", 2)] = mv(K2), mv(torch.linalg.pinv(K2))
    return inputs
