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


# ------------------------------------------------------------------ the four-scale path (mal_loss_multiscale_*), decision-exact
def _lowres(batch, name, s):
    """scale s of a disparity leaf: ``batch["lowres"][name + "_s%d"]`` when the batch brings its own (the reference-generated
    fixtures store fp16-rounded pooled maps), else the pooled copy of scale 0"""
    if s == 0:
        return batch[name]
    lr = batch.get("lowres")
    if lr is not None:
        return lr["%s_s%d" % (name, s)]
    return torch.nn.functional.avg_pool2d(batch[name], 2 ** s)


def ms_build(batch, dev, sclm, double=False):
    """the reference's dict contract for ``sclm`` + 1 scales: lower scales are pooled copies (the shipped decoder emits scale 0
    only); leaves: disp_teacher / disp_student per scale and the four pose vectors"""
    cast = (lambda t: t.double()) if double else (lambda t: t)
    b = {k: (cast(v) if torch.is_tensor(v) and v.dtype == torch.float32 else v) for k, v in batch.items()}  # ("lowres" stays as it is)
    pose = O.transformation_from_parameters if str(dev) == "cpu" else (lambda a, t, inv: None)
    inputs, mono_outputs, outputs, leaves = to_dicts(b, pose, device=None if str(dev) == "cpu" else dev)
    for s in range(1, sclm + 1):
        inputs[("color", 0, s)] = cast(torch.nn.functional.avg_pool2d(batch["color0"], 2 ** s)).to(dev)
        for name, outs in (("disp_teacher", mono_outputs), ("disp_student", outputs)):
            leaf = cast(_lowres(batch, name, s)).to(dev).clone().requires_grad_(True)
            leaves["%s_s%d" % (name, s)] = leaf
            outs[("disp", s)] = leaf
    return inputs, mono_outputs, outputs, leaves


def ms_run_oracle(batch, kw, nt, ns, matching, synth=None, forced=None, double=False, builder=None):
    """process_batch without --distil over scales 0..sclm on the CPU (trainer.py:573-612 calling compute_losses for both
    networks).  ``forced``: {"teacher": [per scale], "student": [per scale], "cmask"} (oracle.mal_oracle.compute_losses)."""
    opt = O.default_opt(**kw)
    sclm = opt.sclm
    cast = (lambda t: t.double()) if double else (lambda t: t)
    # ``builder(dev, double) -> (inputs, mono_outputs, outputs, leaves)``: a test's own dict layout (e.g. one leaf shared by both networks)
    inputs, mono_outputs, outputs, leaves = ms_build(batch, "cpu", sclm, double=double) if builder is None else builder("cpu", double)
    if not matching:
        outputs.pop("lowest_cost", None)
    ft, fs = (None, None) if forced is None else (forced["teacher"], forced["student"])
    has_ins = O.generate_images_pred(opt, inputs, mono_outputs, synth=synth, forced=ft)
    lt = O.compute_losses(opt, inputs, mono_outputs, is_multi=False, has_ins=has_ins, noises=[cast(n.clone()) for n in nt], forced=ft)
    for key in list(mono_outputs.keys()):
        if isinstance(key, tuple) and key[0] in ("depth", "disp"):
            outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
    if matching:  # trainer.py:592-593
        outputs["consistency_mask"] = (outputs["consistency_mask"] * O.compute_matching_mask(outputs) if forced is None
                                       else forced["cmask"].to(outputs["consistency_mask"].dtype))
    O.generate_images_pred(opt, inputs, outputs, is_multi=True, forced=fs)
    ls = O.compute_losses(opt, inputs, outputs, is_multi=True, noises=[cast(n.clone()) for n in ns], forced=fs)
    (lt["loss"] + ls["loss"]).backward()
    res = dict(teacher={k: float(v.detach()) for k, v in lt.items()}, student={k: float(v.detach()) for k, v in ls.items()},
               total=float((lt["loss"] + ls["loss"]).detach()), consistency_mask=outputs["consistency_mask"].detach().numpy(),
               grads={k: (t.grad if t.grad is not None else torch.zeros_like(t)).numpy() for k, t in leaves.items()}, scales=[])
    with torch.no_grad():  # what the tie analysis needs, per scale
        target = inputs[("color", 0, 0)]
        I = torch.cat([O.compute_reprojection_loss(inputs[("color", f, 0)], target, opt.no_ssim) for f in (-1, 1)], 1)
        for s in range(sclm + 1):
            tp = [mono_outputs[("color", f, s)] for f in (-1, 1)]
            if has_ins and opt.temporal:
                tp += [mono_outputs[("syn", f, s)] for f in (-1, 1)]
            sp = [outputs[("color", f, s)] for f in (-1, 1)]
            res["scales"].append(dict(
                t_cands=torch.cat([O.compute_reprojection_loss(c, target, opt.no_ssim) for c in tp], 1).numpy(),
                s_cands=torch.cat([O.compute_reprojection_loss(c, target, opt.no_ssim) for c in sp], 1).numpy(),
                ident=I.min(1, keepdim=True)[0].numpy(), t_preds=[c.numpy() for c in tp], s_preds=[c.numpy() for c in sp],
                t_sample={f: mono_outputs[("sample", f, s)].numpy() for f in (-1, 1)},
                s_sample={f: outputs[("sample", f, s)].numpy() for f in (-1, 1)}))
        res["mono_depth0"] = mono_outputs[("depth", 0, 0)].numpy()
    return res


def _raw_smooth_signs(disp):
    """what the kernels take: the signs of the RAW disparity's first differences at the map's own size (the positive
    1 / (mean + 1e-7) is applied afterwards) -- exact in fp32, so a checker forms them from the inputs"""
    return torch.sign(disp[:, :, :, :-1] - disp[:, :, :, 1:]), torch.sign(disp[:, :, :-1, :] - disp[:, :, 1:, :])


def ms_oracle_decisions(o, batch, nt, sclm, noise_scale=0.00001):
    """the decisions the free-running oracle took in ``o = ms_run_oracle(...)``, in the layout its ``forced=`` expects
    (``noise_scale`` = 0 with --disable_automasking: the identity term is compared without its tie-break noise)"""
    from oracle import aten_restated as AR
    B, _, H, W = batch["color0"].shape
    t = torch.from_numpy
    out = dict(teacher=[], student=[], cmask=t(o["consistency_mask"]))

    def l1_signs(preds, win):
        pred = t(preds[0])
        for i in range(1, len(preds)):
            pred = torch.where(win == i, t(preds[i]), pred)
        return torch.sign(pred - batch["color0"])

    for s in range(sclm + 1):
        sc = o["scales"][s]
        dt, ds = _lowres(batch, "disp_teacher", s), _lowres(batch, "disp_student", s)
        win_t, win_s = t(sc["t_cands"]).argmin(1, keepdim=True), t(sc["s_cands"]).argmin(1, keepdim=True)
        rp_t = t(sc["t_cands"]).min(1, keepdim=True)[0]
        idn = t(sc["ident"]) + nt[s] * noise_scale
        out["teacher"].append(dict(win=win_t, automask=(rp_t <= idn).float(), l1=l1_signs(sc["t_preds"], win_t),
                                   smooth=_smooth_signs(dt), taps={f: AR.taps_of(t(sc["t_sample"][f]), H, W) for f in (-1, 1)}))
        out["student"].append(dict(win=win_s, l1=l1_signs(sc["s_preds"], win_s), smooth=_smooth_signs(ds),
                                   taps={f: AR.taps_of(t(sc["s_sample"][f]), H, W) for f in (-1, 1)}))
    return out


def ms_kernel_decisions(decs, consistency_mask, batch, sclm):
    """``decs`` of mal_amd.step.loss_step_multiscale(want_decisions=True) -> the same layout"""
    def taps(pl):
        pl = pl.long()
        return pl & 0xfff, (pl >> 12) & 0xfff, ((pl >> 24) & 1).bool(), ((pl >> 25) & 1).bool()

    out = dict(teacher=[], student=[], cmask=torch.as_tensor(consistency_mask).cpu().float())
    for s in range(sclm + 1):
        for who, key, name in (("teacher", "dec_teacher", "disp_teacher"), ("student", "dec_student", "disp_student")):
            d = decs[key][s].cpu()
            disp = _lowres(batch, name, s)
            r = dict(win=(d[0].long() & 3)[:, None], taps={-1: taps(d[4]), 1: taps(d[5])},
                     l1=torch.stack([((d[6].long() >> sh) & 3) - 1 for sh in (0, 2, 4)], 1).float(), smooth=_raw_smooth_signs(disp))
            if who == "teacher":
                r["automask"] = ((d[0].long() >> 2) & 1).float()[:, None]
            out[who].append(r)
    return out


def ms_decision_differences(a, b, sclm):
    """per scale and kind: bool (B,1,H,W) map of pixels where two decision sets differ (the smoothness signs are compared by
    the caller at the map's own size)"""
    out = []
    for s in range(sclm + 1):
        d = {"win_t": a["teacher"][s]["win"] != b["teacher"][s]["win"],
             "automask": a["teacher"][s]["automask"] != b["teacher"][s]["automask"],
             "win_s": a["student"][s]["win"] != b["student"][s]["win"]}
        for who, k in (("teacher", "tap_t"), ("student", "tap_s")):
            m = None
            for f in (-1, 1):
                for u, v in zip(a[who][s]["taps"][f], b[who][s]["taps"][f]):
                    m = (u != v) if m is None else (m | (u != v))
            d[k] = m[:, None]
        for who, k in (("teacher", "l1_t"), ("student", "l1_s")):
            d[k] = (a[who][s]["l1"] != b[who][s]["l1"]).any(1, keepdim=True)
        out.append({k: v.numpy() for k, v in d.items()})
    return out


def ms_explicit_route_decisions(opt, inputs, mono_outputs, outputs, nt, batch, sclm):
    """The decisions of the EXPLICIT operator route (MALLossPath with fuse=False: materialising warp, then the materialised-
    candidate kernels) without any instrumented kernel: everything it decides can be re-derived exactly from what it leaves in
    the dicts -- the tap cell / border clip from its own sampling grid ("sample"), the winner and the automask weight by running
    the same deterministic min kernel (ops.photo_fwd) on its own warped images ("color"), the L1 signs from those images (a
    comparison of two fp32 numbers), the smoothness signs from the raw disparities."""
    from mal_amd import _lib as L, loss_utils, ops
    from oracle import aten_restated as AR
    target = inputs[("color", 0, 0)]
    B, _, H, W = target.shape
    no_ssim = bool(getattr(opt, "no_ssim", False))
    flags = L.F_NO_SSIM if no_ssim else 0
    sources = [inputs[("color", f, 0)] for f in (-1, 1)]
    out = dict(teacher=[], student=[], cmask=outputs["consistency_mask"].detach().cpu().float())
    tgt = target.cpu()
    for s in range(sclm + 1):
        for who, outs, name in (("teacher", mono_outputs, "disp_teacher"), ("student", outputs, "disp_student")):
            cands = [outs[("color", f, s)].detach() for f in (-1, 1)]
            if who == "teacher":
                ident = loss_utils.identity_min(target, sources, no_ssim)
                noise = None if getattr(opt, "disable_automasking", False) else nt[s].to(target.device)
                _, am, wt, _ = ops.photo_fwd(target, cands, ident, noise, None, flags | L.F_AUTOMASK)
            else:
                m = torch.ones(B, 1, H, W, dtype=torch.float32, device=target.device)
                if not getattr(opt, "disable_motion_masking", False):
                    m = m * outputs["consistency_mask"].unsqueeze(1)
                if not getattr(opt, "no_matching_augmentation", False):
                    m = m * (1 - outputs["augmentation_mask"][:B])
                _, am, wt, _ = ops.photo_fwd(target, cands, None, None, m.contiguous(), flags)
            win = am.long().cpu()
            pred = torch.where(win == 1, cands[1].cpu(), cands[0].cpu())
            r = dict(win=win, l1=torch.sign(pred - tgt), smooth=_raw_smooth_signs(outs[("disp", s)].detach().cpu()),
                     taps={f: AR.taps_of(outs[("sample", f, s)].detach().cpu(), H, W) for f in (-1, 1)})
            if who == "teacher":
                r["automask"] = wt.cpu()
            out[who].append(r)
    return out
