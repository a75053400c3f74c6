"""N1: the networks of MAL's ``RepDepth`` (manydepth/networks/repdepth.py:23-338) without
torchvision, on PyTorch-ROCm: the convolutions are dense contractions and go to MIOpen/MFMA through
``torch.nn``; what this package adds is the cost volume inside the matching encoder (``mal_amd.costvol``, HIP)
and the pose composition (``mal_amd.layers``).  Module and parameter names follow upstream exactly, so
``model.pth`` / ``encoder.pth`` / ... state dicts interchange (trainer.py:1605-1636, repdepth.py:75-100).

    ResnetEncoder            resnet_encoder.py:362-400   (``encoder.*`` = torchvision's resnet18 layout, fc kept
                                                          and frozen as upstream; multi-image input :15-63)
    ResnetEncoderMatching    resnet_encoder.py:66-119, 264-329 (layer0..4, reduce_conv; cost volume :152-233)
    DepthDecoder             depth_decoder.py:16-68      (monodepth2 decoder, disp at scale 0 only)
    PoseDecoder              pose_decoder.py:13-52
    RepDepth                 repdepth.py:23-338          (teacher, pose net, student; matching augmentation)
"""
from __future__ import annotations

import random
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import costvol
from .layers import transformation_from_parameters


# ------------------------------------------------------------------ ResNet-18 in torchvision's layout
class BasicBlock(nn.Module):
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
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNet18(nn.Module):
    """torchvision.models.resnet18 (random init) with ``num_input_images`` stacked RGB frames
    (resnet_encoder.py:15-41): same attribute names, same initialisation."""

    def __init__(self, num_input_images=1, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2)
        self.layer2 = self._make_layer(128, 2, stride=2)
        self.layer3 = self._make_layer(256, 2, stride=2)
        self.layer4 = self._make_layer(512, 2, stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def _pretrained_unavailable(pretrained):
    if pretrained:
        raise NotImplementedError("ImageNet weights come from torchvision's model zoo upstream "
                                  "(resnet_encoder.py:57-62); load a state dict instead")


class ResnetEncoder(nn.Module):
    """resnet_encoder.py:362-400"""

    def __init__(self, num_layers=18, pretrained=False, num_input_images=1, **kwargs):
        super().__init__()
        if num_layers != 18:
            raise NotImplementedError("MAL uses ResNet-18 everywhere (repdepth.py:41,52)")
        _pretrained_unavailable(pretrained)
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNet18(num_input_images)
        for name, param in self.encoder.named_parameters():
            if "fc" in name:
                param.requires_grad = False

    def forward(self, input_image):
        self.features = []
        x = (input_image - 0.45) / 0.225
        x = self.encoder.bn1(self.encoder.conv1(x))
        self.features.append(self.encoder.relu(x))
        self.features.append(self.encoder.layer1(self.encoder.maxpool(self.features[-1])))
        self.features.append(self.encoder.layer2(self.features[-1]))
        self.features.append(self.encoder.layer3(self.features[-1]))
        self.features.append(self.encoder.layer4(self.features[-1]))
        return self.features


class ResnetEncoderMatching(nn.Module):
    """resnet_encoder.py:66-119 (construction), :121-150 (depth bins), :264-329 (forward).  The cost volume
    (:152-233) and its lowest_cost / confidence_mask (:296-312) run in ``mal_amd.costvol`` (two HIP launches)."""

    def __init__(self, num_layers, pretrained, input_height, input_width, min_depth_bin=0.1, max_depth_bin=20.0,
                 num_depth_bins=96, adaptive_bins=False, depth_binning="linear"):
        super().__init__()
        if num_layers != 18:
            raise NotImplementedError("MAL uses ResNet-18 everywhere")
        _pretrained_unavailable(pretrained)
        self.adaptive_bins = adaptive_bins
        self.depth_binning = depth_binning
        self.set_missing_to_max = True
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.num_depth_bins = num_depth_bins
        self.depth_bins = None
        encoder = ResNet18()
        self.layer0 = nn.Sequential(encoder.conv1, encoder.bn1, encoder.relu)
        self.layer1 = nn.Sequential(encoder.maxpool, encoder.layer1)
        self.layer2 = encoder.layer2
        self.layer3 = encoder.layer3
        self.layer4 = encoder.layer4
        self.reduce_conv = nn.Sequential(nn.Conv2d(self.num_ch_enc[1] + self.num_depth_bins, int(self.num_ch_enc[1]), 3, 1, 1),
                                         nn.ReLU(inplace=True))
        if not adaptive_bins:
            self.compute_depth_bins(min_depth_bin, max_depth_bin)

    def compute_depth_bins(self, min_depth_bin, max_depth_bin):
        lo, hi = float(min_depth_bin), float(max_depth_bin)
        n = self.num_depth_bins
        if self.depth_binning == "inverse":
            bins = torch.from_numpy((1 / np.linspace(1 / hi, 1 / lo, n)[::-1]).copy()).float()
        elif self.depth_binning == "linear":
            bins = torch.linspace(lo, hi, n)
        elif self.depth_binning == "log":
            base, it = np.log(lo), np.log(hi / lo)
            bins = torch.exp(torch.tensor([base + it * i / n for i in range(n)], dtype=torch.float32))
        else:
            raise NotImplementedError(self.depth_binning)
        self.depth_bins = bins

    def feature_extraction(self, image, return_all_feats=False):
        image = (image - 0.45) / 0.225
        feats_0 = self.layer0(image)
        feats_1 = self.layer1(feats_0)
        return [feats_0, feats_1] if return_all_feats else feats_1

    def forward(self, current_image, lookup_images, poses, K, invK, min_depth_bin=None, max_depth_bin=None):
        self.features = self.feature_extraction(current_image, return_all_feats=True)
        current_feats = self.features[-1]
        with torch.no_grad():
            if self.adaptive_bins:
                self.compute_depth_bins(min_depth_bin, max_depth_bin)
            B, Fr, C3, H, W = lookup_images.shape
            lookup_feats = self.feature_extraction(lookup_images.reshape(B * Fr, C3, H, W))
            lookup_feats = lookup_feats.reshape(B, Fr, *lookup_feats.shape[1:])
            cost_volume, lowest_cost, confidence_mask = costvol.cost_volume_outputs(
                current_feats, lookup_feats, poses, K, invK, self.depth_bins, self.set_missing_to_max)
        post_matching_feats = self.reduce_conv(torch.cat([self.features[-1], cost_volume], 1))
        self.features.append(self.layer2(post_matching_feats))
        self.features.append(self.layer3(self.features[-1]))
        self.features.append(self.layer4(self.features[-1]))
        return self.features, lowest_cost, confidence_mask


# ------------------------------------------------------------------ decoders
class Conv3x3(nn.Module):
    """manydepth/layers.py:118-134"""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    """manydepth/layers.py:102-115"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoder(nn.Module):
    """depth_decoder.py:16-68: ``("disp", 0)`` only, whatever ``scales`` says."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels, self.use_skips, self.scales = num_output_channels, use_skips, scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            num_ch_in = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
            num_ch_in = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                num_ch_in += self.num_ch_enc[i - 1]
            self.convs[("upconv", i, 1)] = ConvBlock(num_ch_in, self.num_ch_dec[i])
        self.convs[("dispconv", 0)] = Conv3x3(self.num_ch_dec[0], self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features):
        outputs = {}
        x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            x = [F.interpolate(x, scale_factor=2, mode="nearest")]
            if self.use_skips and i > 0:
                x += [input_features[i - 1]]
            x = self.convs[("upconv", i, 1)](torch.cat(x, 1))
        outputs[("disp", 0)] = self.sigmoid(self.convs[("dispconv", 0)](x))
        return outputs


class PoseDecoder(nn.Module):
    """pose_decoder.py:13-52"""

    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs["squeeze"] = nn.Conv2d(int(num_ch_enc[-1]), 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features):
        cat = torch.cat([self.relu(self.convs["squeeze"](f[-1])) for f in input_features], 1)
        out = cat
        for i in range(3):
            out = self.convs[("pose", i)](out)
            if i != 2:
                out = self.relu(out)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]


# ------------------------------------------------------------------ RepDepth
class RepDepth(nn.Module):
    """repdepth.py:23-338 for the ResNet configuration MAL ships (``pose_cnn`` off): teacher
    (``mono_encoder`` + ``mono_depth``), pose network (``pose_encoder`` + ``pose``), student (``encoder`` with the
    cost volume + ``depth``).  ``forward(inputs, min_depth_bin, max_depth_bin) -> (mono_outputs, outputs)``."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        g = lambda k, d: getattr(opt, k, d)
        self.encoder = ResnetEncoderMatching(g("num_layers", 18), g("weights_init", "scratch") == "pretrained",
                                             input_height=opt.height, input_width=opt.width, adaptive_bins=True,
                                             min_depth_bin=0.1, max_depth_bin=20.0,
                                             depth_binning=g("depth_binning", "linear"),
                                             num_depth_bins=g("num_depth_bins", 96))
        self.depth = DepthDecoder(self.encoder.num_ch_enc, g("scales", [0]))
        self.mono_encoder = ResnetEncoder(18, g("weights_init", "scratch") == "pretrained")
        self.mono_depth = DepthDecoder(self.mono_encoder.num_ch_enc, g("scales", [0]))
        self.pose_encoder = ResnetEncoder(18, g("weights_init", "scratch") == "pretrained", num_input_images=2)
        self.pose = PoseDecoder(self.pose_encoder.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
        self.matching_ids = [0]
        if g("use_future_frame", False):
            self.matching_ids.append(1)
        for idx in range(-1, -1 - g("num_matching_frames", 1), -1):
            self.matching_ids.append(idx)
        self.freeze_tp = False
        self.freeze_pose = False

    def _pose(self, a, b):
        return self.pose([self.pose_encoder(torch.cat([a, b], 1))])

    def predict_poses(self, inputs):
        """repdepth.py:141-235"""
        outputs = {}
        frames = {f: inputs["color_aug", f, 0] for f in self.opt.frame_ids}
        for f_i in self.opt.frame_ids[1:]:
            if f_i == "s":
                continue
            axisangle, translation = self._pose(frames[f_i], frames[0]) if f_i < 0 else self._pose(frames[0], frames[f_i])
            outputs[("axisangle", 0, f_i)] = axisangle
            outputs[("translation", 0, f_i)] = translation
            outputs[("cam_T_cam", 0, f_i)] = transformation_from_parameters(axisangle[:, 0], translation[:, 0],
                                                                            invert=(f_i < 0))
            if getattr(self.opt, "temporal", False):
                outputs[("cam_T_cam", f_i, 0)] = transformation_from_parameters(axisangle[:, 0], translation[:, 0],
                                                                                invert=(f_i > 0))
        # poses for matching, without gradients: 0 -> -1, -1 -> -2, ... chained
        feats = {f: inputs["color_aug", f, 0] for f in self.matching_ids}
        with torch.no_grad():
            for fi in self.matching_ids[1:]:
                if fi < 0:
                    axisangle, translation = self._pose(feats[fi], feats[fi + 1])
                    pose = transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=True)
                    if fi != -1:
                        pose = torch.matmul(pose, inputs[("relative_pose", fi + 1)])
                else:
                    axisangle, translation = self._pose(feats[fi - 1], feats[fi])
                    pose = transformation_from_parameters(axisangle[:, 0], translation[:, 0], invert=False)
                    if fi != 1:
                        pose = torch.matmul(pose, inputs[("relative_pose", fi - 1)])
                missing = (feats[fi].flatten(1).sum(1) == 0).view(-1, 1, 1)  # missing images -> zero pose
                inputs[("relative_pose", fi)] = pose * (~missing).to(pose.dtype)
        return outputs

    def forward(self, inputs, min_depth_bin, max_depth_bin):
        mono_outputs, outputs = {}, {}
        if not self.freeze_tp and not self.freeze_pose:
            pose_pred = self.predict_poses(inputs)
        else:
            with torch.no_grad():
                pose_pred = self.predict_poses(inputs)
        outputs.update(pose_pred)
        mono_outputs.update(pose_pred)
        relative_poses = torch.stack([inputs[("relative_pose", i)] for i in self.matching_ids[1:]], 1)
        lookup_frames = torch.stack([inputs[("color_aug", i, 0)] for i in self.matching_ids[1:]], 1)
        B = len(lookup_frames)
        dev = lookup_frames.device
        # matching augmentation (repdepth.py:268-286): the draws stay on the host RNG as upstream
        augmentation_mask = torch.zeros(B, 1, 1, 1)
        static = torch.zeros(B, dtype=torch.bool)
        dropped = torch.zeros(B, dtype=torch.bool)
        if not getattr(self.opt, "no_matching_augmentation", False):
            for b in range(B):
                r = random.random()
                if r < 0.25:
                    static[b] = True
                elif r < 0.5:
                    dropped[b] = True
                if r < 0.5:
                    augmentation_mask[b] += 1
        if static.any():  # static camera: the lookup frames become the current frame
            sel = static.to(dev).view(B, 1, 1, 1, 1)
            lookup_frames = torch.where(sel, inputs[("color", 0, 0)].unsqueeze(1).expand_as(lookup_frames), lookup_frames)
        if dropped.any():  # missing cost volume: zero poses, the cost volume skips those frames
            relative_poses = relative_poses * (~dropped).to(dev).view(B, 1, 1, 1).to(relative_poses.dtype)
        outputs["augmentation_mask"] = augmentation_mask.to(dev)
        # teacher
        if not self.freeze_tp:
            mono_outputs.update(self.mono_depth(self.mono_encoder(inputs["color_aug", 0, 0])))
        else:
            with torch.no_grad():
                mono_outputs.update(self.mono_depth(self.mono_encoder(inputs["color_aug", 0, 0])))
        for key in list(mono_outputs.keys()):
            if key[0] in ("depth", "disp"):
                outputs[("mono_" + key[0],) + tuple(key[1:])] = mono_outputs[key]
        # student
        features, lowest_cost, confidence_mask = self.encoder(inputs["color_aug", 0, 0], lookup_frames, relative_poses,
                                                              inputs[("K", 2)], inputs[("inv_K", 2)],
                                                              min_depth_bin=min_depth_bin, max_depth_bin=max_depth_bin)
        outputs.update(self.depth(features))
        size = [self.opt.height, self.opt.width]
        outputs["lowest_cost"] = F.interpolate(lowest_cost.unsqueeze(1), size, mode="nearest")[:, 0]
        outputs["consistency_mask"] = F.interpolate(confidence_mask.unsqueeze(1), size, mode="nearest")[:, 0]
        return mono_outputs, outputs
