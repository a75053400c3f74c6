"""CPU: the torchvision-free networks of RepDepth (mal_amd.networks) have upstream's parameter counts
(SURVEY.md 8e hand counts from resnet_encoder.py / depth_decoder.py / pose_decoder.py) and torchvision's /
upstream's state-dict key layout, so checkpoints interchange; the decoders run on CPU."""
import numpy as np
import torch

from mal_amd import harness, networks as N


def count(m, trainable_only=True):
    return sum(p.numel() for p in m.parameters() if p.requires_grad or not trainable_only)


def test_parameter_counts_match_upstream():
    assert count(N.ResnetEncoder(18, False)) == 11_176_512            # resnet18 without its (frozen) fc
    assert count(N.ResnetEncoder(18, False), trainable_only=False) == 11_689_512
    assert count(N.ResnetEncoder(18, False, num_input_images=2)) == 11_185_920   # 6-channel stem
    enc = N.ResnetEncoderMatching(18, False, 192, 640, adaptive_bins=True, num_depth_bins=96)
    assert count(enc) == 11_176_512 + 92_224                          # + reduce_conv (64+96 -> 64, 3x3, bias)
    assert count(N.PoseDecoder(np.array([64, 64, 128, 256, 512]), 1, 2)) == 1_314_572
    assert count(N.DepthDecoder(np.array([64, 64, 128, 256, 512]), [0])) == 3_150_705
    model = N.RepDepth(harness.default_options())
    total = count(model)
    parts = sum(count(m) for m in (model.encoder, model.depth, model.mono_encoder, model.mono_depth, model.pose_encoder, model.pose))
    assert total == parts
    assert total == 41_247_150                                        # "~41.2 M fp32 ~ 165 MB" 


def test_state_dict_layout():
    keys = set(N.ResnetEncoder(18, False).state_dict())
    for k in ("encoder.conv1.weight", "encoder.bn1.running_mean", "encoder.layer1.0.conv1.weight",
              "encoder.layer2.0.downsample.0.weight", "encoder.layer4.1.bn2.num_batches_tracked", "encoder.fc.weight"):
        assert k in keys, k
    keys = set(N.ResnetEncoderMatching(18, False, 192, 640, adaptive_bins=True).state_dict())
    for k in ("layer0.0.weight", "layer0.1.running_var", "layer1.1.0.conv1.weight", "layer2.0.downsample.1.bias",
              "reduce_conv.0.weight", "reduce_conv.0.bias"):
        assert k in keys, k
    keys = set(N.DepthDecoder(np.array([64, 64, 128, 256, 512]), [0]).state_dict())
    assert "decoder.0.conv.conv.weight" in keys and "decoder.10.conv.bias" in keys and len(keys) == 22
    keys = set(N.PoseDecoder(np.array([64, 64, 128, 256, 512]), 1, 2).state_dict())
    assert keys == {"net.%d.%s" % (i, s) for i in range(4) for s in ("weight", "bias")}
    top = {k.split(".")[0] for k in N.RepDepth(harness.default_options()).state_dict()}
    assert top == {"encoder", "depth", "mono_encoder", "mono_depth", "pose_encoder", "pose"}


def test_decoders_run_on_cpu():
    feats = [torch.randn(2, c, 96 // s, 320 // s) for c, s in ((64, 1), (64, 2), (128, 4), (256, 8), (512, 16))]
    disp = N.DepthDecoder(np.array([64, 64, 128, 256, 512]), [0])(feats)[("disp", 0)]
    assert disp.shape == (2, 1, 192, 640) and float(disp.min()) > 0 and float(disp.max()) < 1
    aa, tr = N.PoseDecoder(np.array([64, 64, 128, 256, 512]), 1, 2)([feats])
    assert aa.shape == (2, 2, 1, 3) and tr.shape == (2, 2, 1, 3)
    f = N.ResnetEncoder(18, False)(torch.rand(1, 3, 64, 96))
    assert [t.shape[1] for t in f] == [64, 64, 128, 256, 512] and f[-1].shape[-2:] == (2, 3)


def test_tracker_and_scheduler():
    t = harness.DepthBinTracker(0.1)
    t.update(torch.full((2, 1, 4, 5), 5.0))
    lo, hi = t.compute()
    assert abs(float(lo) - (0.1 * 0.99 + 4.5 * 0.01)) < 1e-6 and abs(float(hi) - (10 * 0.99 + 5.5 * 0.01)) < 1e-6
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], 1e-4)
    s = harness.WarmupStepLRScheduler(opt, 1e-7, 1e-4, warmup_steps=10, decay_steps=5)
    lrs = []
    for _ in range(16):
        s.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert lrs[0] > 1e-7 and abs(lrs[8] - (1e-7 + (1e-4 - 1e-7) / 10 * 9)) < 1e-12 and lrs[-1] < lrs[8]
