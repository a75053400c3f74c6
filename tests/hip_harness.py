"""Shared drivers for the GPU parity tests: run the HIP loss path and the CPU oracle on the
same batch and hand back comparable dictionaries."""
import numpy as np
import torch

from mal_amd.synthetic import to_dicts, fake_image_synthesis
from oracle import mal_oracle as O

LEAVES = ("disp_teacher", "disp_student", "axisangle_m1", "translation_m1", "axisangle_p1", "translation_p1")


def producer_of(batch, device=None):
    """the temporal hint's producer for `batch`: rectangles shifted by torch slicing (``batch["syn_rects"]``: the stand-in
    of the golden vectors, the same code on both sides), or -- ``batch["syn_instances"] = (n_inst, seed)``, the headline as
    bench.py runs it -- dyn_utils.image_synthesis itself driven by the stand-in segmenter / matcher of
    mal_amd.synthetic.instance_stub: the HIP kernels (mal_amd.dyn_utils, sparse syn buffers + region map) on a device, the
    CPU restatement (oracle.dyn_oracle.image_synthesis, pinned to the reference's own function) for the oracle."""
    if "syn_instances" in batch:
        from mal_amd.synthetic import instance_stub
        n_inst, seed = batch["syn_instances"]
        B, _, H, W = batch["color0"].shape
        ins_model, matcher = instance_stub(B, H, W, n_inst=n_inst, seed=seed, device="cpu" if device is None else device)
        if device is None:
            from oracle import dyn_oracle
            return lambda inputs, outputs, scale: dyn_oracle.image_synthesis(inputs, outputs, scale, 0.5, ins_model, matcher)
        from mal_amd import dyn_utils
        return lambda inputs, outputs, scale: dyn_utils.image_synthesis(inputs, outputs, scale, 0.5, ins_model, matcher)
    return fake_image_synthesis(batch["syn_rects"]) if "syn_rects" in batch else None


def run_oracle(batch, opt_kw, n0, n1, w_list=(0.7, 0.3), forced=None):
    """``forced``: the per-pixel decisions to take instead of re-deciding them (oracle.mal_oracle.mal_loss_step)."""
    B, _, H, W = batch["color0"].shape
    opt = O.default_opt(height=H, width=W, batch_size=B, **opt_kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, O.transformation_from_parameters)
    synth = producer_of(batch)
    losses, loss_list, mono_losses, mono_reproj, ens = O.mal_loss_step(
        opt, inputs, mono_outputs, outputs, n0.clone(), n1.clone(), list(w_list), synth=synth, forced=forced)
    final = B * (w_list[0] * loss_list[0] + w_list[1] * loss_list[1]) if opt.loss_blc else losses["loss"]
    final.backward()
    # maps the tests need for tie analysis
    with torch.no_grad():
        target = inputs[("color", 0, 0)]
        multi_pred = [outputs[("color", f, 0)] for f in (-1, 1)]
        if ("syn", -1, 0) in outputs and opt.main_temporal:
            multi_pred += [outputs[("syn", f, 0)] for f in (-1, 1)]
        R = torch.cat([O.compute_reprojection_loss(c, target) for c in multi_pred], 1)
        mono_pred = [mono_outputs[("color", f, 0)] for f in (-1, 1)]
        if ("syn", -1, 0) in mono_outputs and opt.temporal:  # the teacher's four candidates (loss_utils.py:79-90)
            mono_pred += [mono_outputs[("syn", f, 0)] for f in (-1, 1)]
        Rm = torch.cat([O.compute_reprojection_loss(c, target) for c in mono_pred], 1)
        I = torch.cat([O.compute_reprojection_loss(inputs[("color", f, 0)], target) for f in (-1, 1)], 1)
    return dict(final=final.item(), losses={k: v.item() for k, v in losses.items()},
                mono_losses={k: v.item() for k, v in mono_losses.items()},
                loss_list=None if loss_list is None else [l.item() for l in loss_list],
                mono_reproj=mono_reproj.detach().numpy(), ens=None if ens is None else ens.detach().numpy(),
                multi_cands=R.numpy(), mono_cands=Rm.numpy(), ident=I.min(1, keepdim=True)[0].numpy(),
                multi_depth=outputs[("depth", 0, 0)].detach().numpy(),
                mono_depth=mono_outputs[("depth", 0, 0)].detach().numpy(),
                consistency_mask=outputs["consistency_mask"].numpy(),
                cons_target=outputs["consistency_target/0"].numpy(),
                mono_color={f: mono_outputs[("color", f, 0)].detach().numpy() for f in (-1, 1)},
                mono_preds=[c.detach().numpy() for c in mono_pred],
                multi_preds=[c.detach().numpy() for c in multi_pred],
                multi_color={f: outputs[("color", f, 0)].detach().numpy() for f in (-1, 1)},
                mono_sample={f: mono_outputs[("sample", f, 0)].detach().numpy() for f in (-1, 1)},
                multi_sample={f: outputs[("sample", f, 0)].detach().numpy() for f in (-1, 1)},
                grads={k: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy() for k, t in leaves.items()})


def run_hip(batch, opt_kw, n0, n1, fuse=True, w_list=(0.7, 0.3), device="cuda:0"):
    from mal_amd import layers, trainer
    B, _, H, W = batch["color0"].shape
    dev = torch.device(device)
    opt = trainer.default_options(height=H, width=W, batch_size=B, **opt_kw)
    inputs, mono_outputs, outputs, leaves = to_dicts(batch, layers.transformation_from_parameters, device=dev)
    synth = producer_of(batch, dev)
    lp = trainer.LossPath(opt, fuse=fuse, image_synthesis=synth)
    lp.w_list = list(w_list)
    # the reference draws the two noise tensors inside the loss functions; hand them in
    from mal_amd import loss_utils
    keep = loss_utils.draw_noise
    seq = [n0.to(dev), n1.to(dev)]
    loss_utils.draw_noise = lambda shape, device: seq.pop(0)
    from mal_amd import config
    old = config.noise_source
    config.noise_source = "given"
    stash = {}
    keep_mono, keep_ens = loss_utils.compute_mono_losses, lp.generate_images_pred_ensemble

    def mono_wrap(*a, **k):
        l, mr = keep_mono(*a, **k)
        stash["mono_reproj"] = mr.detach().cpu().numpy()
        return l, mr

    def ens_wrap(*a, **k):
        e = keep_ens(*a, **k)
        stash["ens"] = e.detach().cpu().numpy()
        return e

    loss_utils.compute_mono_losses = mono_wrap
    lp.generate_images_pred_ensemble = ens_wrap
    try:
        outs, losses, loss_list = lp.compute_batch_losses(inputs, mono_outputs, outputs)
    finally:
        loss_utils.draw_noise = keep
        loss_utils.compute_mono_losses = keep_mono
        config.noise_source = old
    final = B * (w_list[0] * loss_list[0] + w_list[1] * loss_list[1]) if opt.loss_blc else losses["loss"]
    final.backward()
    torch.cuda.synchronize()
    res = dict(final=final.item(), losses={k: float(v.detach()) for k, v in losses.items()},
               loss_list=None if loss_list is None else [float(l.detach()) for l in loss_list],
               multi_depth=outputs[("depth", 0, 0)].detach().cpu().numpy(),
               mono_depth=mono_outputs[("depth", 0, 0)].detach().cpu().numpy(),
               consistency_mask=outputs["consistency_mask"].cpu().numpy(),
               grads={k: (t.grad if t.grad is not None else torch.zeros_like(t)).cpu().numpy()
                      for k, t in leaves.items()},
               outputs=outputs, mono_outputs=mono_outputs, lp=lp, mono_reproj=stash.get("mono_reproj"),
               ens=stash.get("ens"))
    if "consistency_target/0" in outputs:
        res["cons_target"] = outputs["consistency_target/0"].cpu().numpy()
    return res


# ------------------------------------------------------------------ decision-exact parity (tests/test_gpu_decisions.py)
DEC_KINDS = ("win_t", "automask", "win_s", "distil", "cmask", "tap_t", "tap_s", "smooth_t", "smooth_s", "l1_t", "l1_s")


def _smooth_signs(disp):
    """signs of the first differences of the mean-normalised disparity (layers.py:210-223, loss_utils.py:119-121)"""
    n = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
    return torch.sign(n[:, :, :, :-1] - n[:, :, :, 1:]), torch.sign(n[:, :, :-1, :] - n[:, :, 1:, :])


def oracle_decisions(o, batch, n0, no_ens=False):
    """The decisions the free-running oracle took in ``o = run_oracle(...)``, in the layout ``forced=`` expects."""
    from oracle import aten_restated as AR
    B, _, H, W = batch["color0"].shape
    t = torch.from_numpy
    idn = t(o["ident"]) + n0 * 0.00001
    rp_t = t(o["mono_cands"]).min(1, keepdim=True)[0]

    def l1_signs(preds, win):  # sign(pred - target) of the winning candidate, per channel
        pred = t(preds[0])
        for i in range(1, len(preds)):
            pred = torch.where(win == i, t(preds[i]), pred)
        return torch.sign(pred - batch["color0"])

    win_t, win_s = t(o["mono_cands"]).argmin(1, keepdim=True), t(o["multi_cands"]).argmin(1, keepdim=True)
    teacher = dict(win=win_t, automask=(rp_t <= idn).float(), l1=l1_signs(o["mono_preds"], win_t),
                   smooth=_smooth_signs(batch["disp_teacher"]),
                   taps={f: AR.taps_of(t(o["mono_sample"][f]), H, W) for f in (-1, 1)})
    rp_s = t(o["multi_cands"]).min(1, keepdim=True)[0]
    trio = [t(o["mono_reproj"])] + ([] if no_ens or o["ens"] is None else [t(o["ens"])]) + [rp_s]
    idx = torch.cat(trio, 1).argmin(1, keepdim=True)
    if len(trio) == 2:
        idx = idx * 2  # numbered as the kernels do: 0 teacher, 2 student
    student = dict(win=win_s, distil=idx, l1=l1_signs(o["multi_preds"], win_s),
                   smooth=_smooth_signs(batch["disp_student"]),
                   taps={f: AR.taps_of(t(o["multi_sample"][f]), H, W) for f in (-1, 1)})
    return dict(teacher=teacher, student=student, cmask=t(o["consistency_mask"]))


def kernel_decisions(maps):
    """``maps`` of mal_amd.step.loss_step(want_decisions=True) -> the same layout (include/mal_hip.h MAL_DEC_*)."""
    def taps(pl):
        pl = pl.long()
        return pl & 0xfff, (pl >> 12) & 0xfff, ((pl >> 24) & 1).bool(), ((pl >> 25) & 1).bool()

    def one(d, student):
        d = torch.as_tensor(d).cpu()
        r = dict(win=(d[0].long() & 3)[:, None], smooth=((d[2][:, None, :, :-1] - 1).float(), (d[3][:, None, :-1, :] - 1).float()),
                 taps={-1: taps(d[4]), 1: taps(d[5])},
                 l1=torch.stack([((d[6].long() >> s) & 3) - 1 for s in (0, 2, 4)], 1).float())
        if student:
            r["distil"] = d[1].long()[:, None]
        else:
            r["automask"] = ((d[0].long() >> 2) & 1).float()[:, None]
        return r
    return dict(teacher=one(maps["dec_teacher"], False), student=one(maps["dec_student"], True),
                cmask=torch.as_tensor(maps["consistency_mask"]).cpu().float())


def decision_differences(a, b):
    """per kind: bool (B,1,H,W)-broadcastable map of pixels where two decision sets differ"""
    d = {}
    d["win_t"] = a["teacher"]["win"] != b["teacher"]["win"]
    d["automask"] = a["teacher"]["automask"] != b["teacher"]["automask"]
    d["win_s"] = a["student"]["win"] != b["student"]["win"]
    d["distil"] = a["student"]["distil"] != b["student"]["distil"]
    d["cmask"] = (a["cmask"] != b["cmask"])[:, None]
    for who, k in (("teacher", "tap_t"), ("student", "tap_s")):
        m = None
        for f in (-1, 1):
            for u, v in zip(a[who]["taps"][f], b[who]["taps"][f]):
                m = (u != v) if m is None else (m | (u != v))
        d[k] = m[:, None]
    for who, k in (("teacher", "l1_t"), ("student", "l1_s")):
        d[k] = (a[who]["l1"] != b[who]["l1"]).any(1, keepdim=True)
    for who, k in (("teacher", "smooth_t"), ("student", "smooth_s")):
        (ax, ay), (bx, by) = a[who]["smooth"], b[who]["smooth"]
        m = torch.zeros(ax.shape[0], 1, ax.shape[2], ax.shape[3] + 1, dtype=torch.bool)
        m[..., :, :-1] |= ax != bx
        m[..., :-1, :] |= ay != by
        d[k] = m
    return {k: v.numpy() for k, v in d.items()}


def dilate3(mask):
    """3x3 dilation of a (B,1,H,W) bool array (a flipped pixel moves its neighbours' gradients)."""
    m = np.pad(mask, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros_like(mask)
    H, W = mask.shape[-2:]
    for dy in range(3):
        for dx in range(3):
            out |= m[..., dy:dy + H, dx:dx + W]
    return out


def automask_tie_allowance(o, n0, rel=3e-5):
    """The automask `rp <= identity + 1e-5*noise` (loss_utils.py:27-44) compares two fp32 values: a pixel within
    rounding distance of the threshold may fall on either side.  Returns (allow, renorm, any): flipping pixel i
    moves the masked mean sum(rp*m)/sum(m) by at most (rp_i + mean)/sum(m) [allow, summed over such pixels],
    rescales every teacher-pass gradient by 1/sum(m) -> 1/(sum(m) +- 1) [renorm], and adds or removes that
    pixel's whole contribution to the summed (pose) gradients [any]."""
    idn = o["ident"] + n0.numpy() * np.float32(1e-5)
    amb = np.abs(o["mono_reproj"] - idn) <= rel * np.maximum(np.abs(idn), 1e-3)
    m = o["mono_reproj"] <= idn
    mean = float((o["mono_reproj"] * m).sum() / max(int(m.sum()), 1))
    allow = float(((o["mono_reproj"] + mean) * amb).sum() / max(int(m.sum()), 1))
    return allow, float(amb.sum()) / max(int(m.sum()), 1), bool(amb.any())


def smooth_sign_ambiguous(disp, rel=4e-7):
    """(B,1,H,W) bool: a neighbour's disparity is within a few ulp -- the sign of d(mean-normalised disp) in
    get_smooth_loss's gradient (layers.py:210-223) then depends on how the normalisation rounds (upstream divides
    each pixel by mean+1e-7 first; the kernels take the difference of the raw values and scale afterwards)."""
    d = disp.astype(np.float64)
    amb = np.zeros(d.shape, dtype=bool)
    tol = rel * np.abs(d)
    dx = np.abs(d[..., :, 1:] - d[..., :, :-1])
    dy = np.abs(d[..., 1:, :] - d[..., :-1, :])
    ex = (dx <= np.maximum(tol[..., :, 1:], tol[..., :, :-1])) & (dx > 0)
    ey = (dy <= np.maximum(tol[..., 1:, :], tol[..., :-1, :])) & (dy > 0)
    amb[..., :, 1:] |= ex
    amb[..., :, :-1] |= ex
    amb[..., 1:, :] |= ey
    amb[..., :-1, :] |= ey
    return amb


def near_tie(maps, tol, distinct=False):
    """pixels where the smallest two of the stacked (B,K,H,W) maps are within tol (relative).  ``distinct``: candidates
    that are EXACTLY equal count as one (the temporal hint's synthesised image equals the warped one wherever no
    instance moved: whichever of the two is reported as the winner, the gradient reaches the same pixels)."""
    s = np.sort(maps, axis=1)
    if not distinct:
        return (np.abs(s[:, 1:2] - s[:, 0:1]) <= tol * np.maximum(np.abs(s[:, 0:1]), 1e-3))
    gap = np.full(s[:, 0:1].shape, np.inf)
    for k in range(s.shape[1] - 1, 0, -1):
        g = s[:, k:k + 1] - s[:, 0:1]
        gap = np.where(g > 0, g, gap)
    return gap <= tol * np.maximum(np.abs(s[:, 0:1]), 1e-3)


def sample_ambiguous(sample, H, W, tol=5e-4):
    """(B,1,H,W) bool: sampling position within tol pixels of an integer (the bilinear taps,
    hence d/d(u,v), switch there) or of the border clip (gradient gate)."""
    amb = np.zeros(sample[-1].shape[:3], dtype=bool)
    for f in (-1, 1):
        g = sample[f].astype(np.float64)
        ix = (g[..., 0] + 1) / 2 * (W - 1)
        iy = (g[..., 1] + 1) / 2 * (H - 1)
        for v, hi in ((ix, W - 1), (iy, H - 1)):
            amb |= np.abs(v - np.round(v)) <= tol
            amb |= (np.abs(v) <= tol) | (np.abs(v - hi) <= tol)
    return amb[:, None]


def oracle_fp64_grads(batch, opt_kw, n0, n1):
    b64 = {k: (v.double() if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in batch.items()}
    return run_oracle(b64, opt_kw, n0.double(), n1.double())["grads"]
