"""Golden vectors for the networks around the loss path.  TEST INFRASTRUCTURE ONLY; authoring
container only:

    python -m oracle.gen_golden_net

Imports the reference's OWN ``manydepth.networks`` package (``RepDepth``, ``ResnetEncoderMatching``, ``ResnetEncoder``,
``DepthDecoder``, ``PoseDecoder``; manydepth/networks/*.py) and runs it on the CPU.  The package needs torchvision, which
this image lacks and which is not part of the reference: an inert stand-in supplies what the reference takes from it -- a
plain ResNet (``models.ResNet`` with ``_make_layer``, ``models.resnet18``, ``models.resnet.BasicBlock``) in torchvision's
attribute layout; it carries no reference code and no learned weights.  Every parameter and buffer of the reference
model is filled from a generator seeded by its state-dict NAME (tests/net_weights.py), so the tests rebuild the same
weights inside ``mal_amd.networks`` without a 165 MB fixture: a wrong key, shape, skip connection, frame order, pose
sign or detach would show as a mismatch.

Writes tests/golden/net_parts_*.npz (encoders / decoders: outputs and input gradients) and net_repdepth_*.npz
(``RepDepth.forward`` incl. ``predict_poses``, the matching augmentation under a fixed ``random.seed``, a missing
lookup frame, train and eval mode: outputs and the gradient w.r.t. the input images).
"""
from __future__ import annotations

import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---------------------------------------------------------------- the torchvision stand-in (not reference code)
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class _ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def install_torchvision_stand_in():
    if "torchvision" in sys.modules and not getattr(sys.modules["torchvision"], "_mal_stand_in", False):
        return  # a real torchvision: use it
    tv, models, resnet = types.ModuleType("torchvision"), types.ModuleType("torchvision.models"), types.ModuleType("torchvision.models.resnet")
    tv._mal_stand_in = True
    resnet.BasicBlock, resnet.Bottleneck, resnet.model_urls = _BasicBlock, _BasicBlock, {}
    models.ResNet = _ResNet

    def resnet18(pretrained=False, **kw):
        assert not pretrained, "no network: random init only"
        return _ResNet(_BasicBlock, [2, 2, 2, 2])

    def unavailable(*a, **k):
        raise NotImplementedError("only resnet18 exists in the stand-in")

    models.resnet18 = resnet18
    models.resnet34 = models.resnet50 = models.resnet101 = models.resnet152 = unavailable
    models.resnet, tv.models = resnet, models
    sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.models.resnet": resnet})


def import_reference_networks():
    install_torchvision_stand_in()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    stale = sys.modules.get("manydepth.networks")
    if stale is not None and not hasattr(stale, "RepDepth"):  # a bare namespace left by another generator
        for k in [k for k in sys.modules if k.startswith("manydepth.networks")]:
            del sys.modules[k]
    import manydepth.networks as RN
    return RN


# ---------------------------------------------------------------- shared case definitions (the tests rebuild the inputs)
NUM_CH_ENC = np.array([64, 64, 128, 256, 512])


def parts_inputs(seed=3, B=2, H=64, W=96):
    g = torch.Generator().manual_seed(seed)
    feats = [torch.randn(B, c, max(H // s, 1), max(W // s, 1), generator=g).relu() for c, s in
             zip(NUM_CH_ENC, (2, 4, 8, 16, 32))]
    pose_feat = torch.randn(B, 512, 2, 3, generator=g).relu()
    img1 = torch.rand(B, 3, H, W, generator=g)
    img2 = torch.rand(B, 6, H, W, generator=g)
    cot = {"disp": torch.randn(B, 1, H, W, generator=g), "aa": torch.randn(B, 2, 1, 3, generator=g),
           "tr": torch.randn(B, 2, 1, 3, generator=g),
           "f1": [torch.randn(B, c, max(H // s, 1), max(W // s, 1), generator=g) for c, s in zip(NUM_CH_ENC, (2, 4, 8, 16, 32))],
           "f2": [torch.randn(B, c, max(H // s, 1), max(W // s, 1), generator=g) for c, s in zip(NUM_CH_ENC, (2, 4, 8, 16, 32))]}
    return feats, pose_feat, img1, img2, cot


def repdepth_options(H, W, **kw):
    o = dict(height=H, width=W, num_layers=18, weights_init="scratch", depth_binning="linear", num_depth_bins=96,
             scales=[0], pose_cnn=False, use_future_frame=False, num_matching_frames=1, dc=False, frame_ids=[0, -1, 1],
             temporal=True, no_matching_augmentation=False, batch_size=4)
    o.update(kw)
    return types.SimpleNamespace(**o)


def repdepth_inputs(B, H, W, seed, missing_sample=None):
    """the dictionary RepDepth.forward reads (mono_dataset.py:150-200): colour frames, their augmented versions (a fixed
    brightness / contrast change, so that "color" and "color_aug" cannot be mixed up unnoticed), intrinsics at scale 2"""
    from tests.net_weights import seeded_images
    from mal_amd.synthetic import make_batch
    u8 = seeded_images(B, H, W, seed)
    inputs = {}
    for f, t in u8.items():
        c = t.float() / 255.0
        inputs[("color", f, 0)] = c
        inputs[("color_aug", f, 0)] = (0.9 * c + 0.04).clamp(0, 1)
    if missing_sample is not None:  # a missing lookup frame: all zeros -> zero pose (repdepth.py:222-226)
        inputs[("color_aug", -1, 0)][missing_sample] = 0.0
        inputs[("color", -1, 0)][missing_sample] = 0.0
    bt = make_batch(B, H, W, seed=seed)
    K2 = bt["K"].clone()
    K2[:, 0] /= 4
    K2[:, 1] /= 4
    inputs[("K", 2)], inputs[("inv_K", 2)] = K2, torch.linalg.pinv(K2)
    return inputs, u8


def repdepth_cotangents(B, H, W, seed):
    g = torch.Generator().manual_seed(seed + 1000)
    return {"disp": torch.randn(B, 1, H, W, generator=g), "mono_disp": torch.randn(B, 1, H, W, generator=g),
            "aa": torch.randn(B, 2, 1, 3, generator=g), "tr": torch.randn(B, 2, 1, 3, generator=g)}


def repdepth_scalar(mono_outputs, outputs, cot):
    """the scalar whose gradient w.r.t. the input images the fixture stores: touches both decoders and both poses"""
    s = (outputs[("disp", 0)] * cot["disp"]).sum() + (mono_outputs[("disp", 0)] * cot["mono_disp"]).sum()
    for f in (-1, 1):
        s = s + 100.0 * ((mono_outputs[("axisangle", 0, f)] * cot["aa"]).sum() + (mono_outputs[("translation", 0, f)] * cot["tr"]).sum())
        s = s + outputs[("cam_T_cam", 0, f)].sum() + outputs[("cam_T_cam", f, 0)].sum()
    return s


def run_repdepth(model, inputs, cot, aug_seed, train, min_bin=0.4, max_bin=9.0):
    leaves = {}
    inp = dict(inputs)
    for f in (0, -1, 1):
        leaves[f] = inputs[("color_aug", f, 0)].clone().requires_grad_(True)
        inp[("color_aug", f, 0)] = leaves[f]
    model.train(train)
    random.seed(aug_seed)
    mono_outputs, outputs = model(inp, torch.tensor(min_bin), torch.tensor(max_bin))
    s = repdepth_scalar(mono_outputs, outputs, cot)
    s.backward()
    d = {"out/disp": outputs[("disp", 0)], "out/mono_disp": mono_outputs[("disp", 0)], "out/mono_disp_in_outputs": outputs[("mono_disp", 0)],
         "out/lowest_cost": outputs["lowest_cost"], "out/consistency_mask": outputs["consistency_mask"],
         "out/augmentation_mask": outputs["augmentation_mask"], "out/relative_pose_m1": inp[("relative_pose", -1)],
         "out/scalar": s.detach().reshape(1)}
    for f in (-1, 1):
        d["out/axisangle_%d" % f] = mono_outputs[("axisangle", 0, f)]
        d["out/translation_%d" % f] = mono_outputs[("translation", 0, f)]
        d["out/cam_T_cam_0_%d" % f] = outputs[("cam_T_cam", 0, f)]
        d["out/cam_T_cam_%d_0" % f] = outputs[("cam_T_cam", f, 0)]
    for f in (0, -1, 1):
        d["grad/color_aug_%d" % f] = leaves[f].grad if leaves[f].grad is not None else torch.zeros_like(leaves[f])
    return {k: np.asarray(torch.as_tensor(v).detach().float().cpu().numpy()) for k, v in d.items()}


def main():
    from tests.net_weights import named_fill_
    torch.set_num_threads(1)  # one summation order, whatever the host
    RN = import_reference_networks()
    os.makedirs(OUT, exist_ok=True)

    # ---- parts: decoders and the plain encoders, outputs + input gradients
    feats, pose_feat, img1, img2, cot = parts_inputs()
    d = {}
    dec = RN.DepthDecoder(NUM_CH_ENC, [0])
    named_fill_(dec, seed=1)
    fl = [f.clone().requires_grad_(True) for f in feats]
    disp = dec(fl)[("disp", 0)]
    (disp * cot["disp"]).sum().backward()
    d["depth_decoder/disp"] = disp.detach().numpy()
    for i in (0, 4):  # the shallowest skip and the bottleneck (the others would only add bytes)
        d["depth_decoder/grad_feat%d" % i] = fl[i].grad.numpy()
    pd = RN.PoseDecoder(NUM_CH_ENC, num_input_features=1, num_frames_to_predict_for=2)
    named_fill_(pd, seed=2)
    pf = pose_feat.clone().requires_grad_(True)
    aa, tr = pd([[pf]])
    ((aa * cot["aa"]).sum() + (tr * cot["tr"]).sum()).backward()
    d["pose_decoder/axisangle"], d["pose_decoder/translation"], d["pose_decoder/grad_feat"] = aa.detach().numpy(), tr.detach().numpy(), pf.grad.numpy()
    for tag, n_img, img, ck in (("encoder1", 1, img1, "f1"), ("encoder2", 2, img2, "f2")):
        enc = RN.ResnetEncoder(18, False, num_input_images=n_img)
        named_fill_(enc, seed=3)
        for mode in ("train", "eval"):
            enc.train(mode == "train")
            x = img.clone().requires_grad_(True)
            fs = enc(x)
            sum((f * c).sum() for f, c in zip(fs, cot[ck])).backward()
            for i in (1, 4):
                d["%s/%s/feat%d" % (tag, mode, i)] = fs[i].detach().numpy()
            d["%s/%s/grad_image" % (tag, mode)] = x.grad.numpy()
        d["%s/trainable" % tag] = np.array(sum(p.numel() for p in enc.parameters() if p.requires_grad))
    np.savez_compressed(os.path.join(OUT, "net_parts_b2_64x96.npz"), **d)
    print("net_parts_b2_64x96", {k: v.shape for k, v in list(d.items())[:4]}, "...")

    # ---- RepDepth.forward: a seed whose four random draws cover static camera, dropped cost volume and no augmentation
    B, H, W, seed = 4, 64, 96, 5
    aug_seed = next(s for s in range(1000) if _covers(s, B))
    inputs, u8 = repdepth_inputs(B, H, W, seed, missing_sample=B - 1)
    opt = repdepth_options(H, W, batch_size=B)
    model = RN.RepDepth(opt)
    n = named_fill_(model, seed=4)
    cotr = repdepth_cotangents(B, H, W, seed)
    out = {"in/aug_seed": np.array(aug_seed), "in/seed": np.array(seed), "in/weights_written": np.array(n),
           "in/state_dict_keys": np.array(sorted(model.state_dict().keys())),
           "in/trainable": np.array(sum(p.numel() for p in model.parameters() if p.requires_grad))}
    for f, t in u8.items():
        out["in/color_u8_%d" % f] = t.numpy()
    import copy
    sd0 = copy.deepcopy(model.state_dict())
    for mode in ("train", "eval"):
        model.load_state_dict(sd0, strict=False)  # train mode moves the running statistics (forward adds geometry buffers)
        r = run_repdepth(model, inputs, cotr, aug_seed, train=(mode == "train"))
        for k, v in r.items():
            out["%s/%s" % (mode, k)] = v
        print("net_repdepth", mode, "augmentation", r["out/augmentation_mask"].ravel(), "scalar", float(r["out/scalar"][0]),
              "consistency", float(r["out/consistency_mask"].mean()))
    np.savez_compressed(os.path.join(OUT, "net_repdepth_b4_64x96.npz"), **out)
    print("wrote net_repdepth_b4_64x96:", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def _covers(s, B):
    random.seed(s)
    r = [random.random() for _ in range(B)]
    return any(x < 0.25 for x in r[:-1]) and any(0.25 <= x < 0.5 for x in r[:-1]) and any(x >= 0.5 for x in r[:-1])


if __name__ == "__main__":
    main()
