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


# ---------------------------------------------------------------- N1 parity: the reference's own networks (fixtures)
# tests/golden/net_*.npz come from /root/reference/manydepth/networks run on the CPU (oracle/gen_golden_net.py); weights
# are rebuilt here from the state-dict NAMES (tests/net_weights.py), so keys and shapes are part of what is compared.
def _close(a, b, what, rel=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(np.abs(b).max()), 1e-30)
    err = float(np.abs(a - b).max()) / scale
    assert err <= rel, (what, err)


def test_decoders_and_encoders_equal_the_reference_networks(golden_dir):
    import os
    from oracle.gen_golden_net import parts_inputs, NUM_CH_ENC
    from tests.net_weights import named_fill_
    torch.set_num_threads(1)
    z = np.load(os.path.join(golden_dir, "net_parts_b2_64x96.npz"))
    feats, pose_feat, img1, img2, cot = parts_inputs()
    dec = N.DepthDecoder(NUM_CH_ENC, [0])
    named_fill_(dec, seed=1)
    fl = [f.clone().requires_grad_(True) for f in feats]
    disp = dec(fl)[("disp", 0)]
    (disp * cot["disp"]).sum().backward()
    _close(disp.detach(), z["depth_decoder/disp"], "depth decoder disp")
    for i in (0, 4):
        _close(fl[i].grad, z["depth_decoder/grad_feat%d" % i], "depth decoder grad feat%d" % i)
    pd = N.PoseDecoder(NUM_CH_ENC, num_input_features=1, num_frames_to_predict_for=2)
    named_fill_(pd, seed=2)
    pf = pose_feat.clone().requires_grad_(True)
    aa, tr = pd([[pf]])
    ((aa * cot["aa"]).sum() + (tr * cot["tr"]).sum()).backward()
    _close(aa.detach(), z["pose_decoder/axisangle"], "axisangle")
    _close(tr.detach(), z["pose_decoder/translation"], "translation")
    _close(pf.grad, z["pose_decoder/grad_feat"], "pose decoder grad")
    for tag, n_img, img, ck in (("encoder1", 1, img1, "f1"), ("encoder2", 2, img2, "f2")):
        enc = N.ResnetEncoder(18, False, num_input_images=n_img)
        named_fill_(enc, seed=3)
        assert sum(p.numel() for p in enc.parameters() if p.requires_grad) == int(z["%s/trainable" % tag])
        for mode in ("train", "eval"):
            enc.train(mode == "train")
            x = img.clone().requires_grad_(True)
            fs = enc(x)
            sum((f * c).sum() for f, c in zip(fs, cot[ck])).backward()
            for i in (1, 4):
                _close(fs[i].detach(), z["%s/%s/feat%d" % (tag, mode, i)], "%s %s feat%d" % (tag, mode, i))
            _close(x.grad, z["%s/%s/grad_image" % (tag, mode)], "%s %s grad image" % (tag, mode), rel=1e-5)


def test_repdepth_glue_equals_the_reference_on_cpu(golden_dir, monkeypatch):
    """RepDepth.forward / predict_poses (repdepth.py:141-338) and ResnetEncoderMatching.forward (resnet_encoder.py:264-329)
    against the reference's own, on the CPU: the two HIP pieces inside mal_amd.networks (cost volume, pose composition)
    are swapped for the CPU checkers here -- tests/test_gpu_networks.py runs the same fixture with the HIP pieces in place.
    Covers: frame order and pose signs, cam_T_cam (f,0) of --temporal, relative-pose chaining without gradient, the
    missing-frame zero pose, static-camera / dropped-cost-volume augmentation under the fixture's random.seed, color vs
    color_aug, mono_* aliases, nearest upsampling of lowest_cost / consistency_mask, train and eval mode."""
    import os
    from oracle import costvol_oracle as CO, mal_oracle as O
    from oracle.gen_golden_net import repdepth_inputs, repdepth_options, repdepth_cotangents, run_repdepth
    from tests.net_weights import named_fill_
    torch.set_num_threads(1)
    z = np.load(os.path.join(golden_dir, "net_repdepth_b4_64x96.npz"))
    B, H, W, seed = 4, 64, 96, int(z["in/seed"])

    def cpu_cost_volume_outputs(cur, look, poses, K, invK, bins, set_missing_to_max=True):
        cv, miss = CO.match_features(cur, look, poses, K, invK, torch.as_tensor(bins), set_missing_to_max)
        return CO.encoder_outputs(cv, miss, torch.as_tensor(bins))

    monkeypatch.setattr(N.costvol, "cost_volume_outputs", cpu_cost_volume_outputs)
    monkeypatch.setattr(N, "transformation_from_parameters", lambda a, t, invert=False: O.transformation_from_parameters(a, t, invert))
    inputs, u8 = repdepth_inputs(B, H, W, seed, missing_sample=B - 1)
    for f, t in u8.items():
        assert np.array_equal(t.numpy(), z["in/color_u8_%d" % f])
    model = N.RepDepth(repdepth_options(H, W, batch_size=B))
    assert sorted(model.state_dict().keys()) == [str(k) for k in z["in/state_dict_keys"] if "backprojector" not in str(k) and "projector" not in str(k)]
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(z["in/trainable"])
    n = named_fill_(model, seed=4)
    assert n == int(z["in/weights_written"])
    import copy
    sd0 = copy.deepcopy(model.state_dict())
    cot = repdepth_cotangents(B, H, W, seed)
    for mode in ("train", "eval"):
        model.load_state_dict(sd0)
        r = run_repdepth(model, inputs, cot, int(z["in/aug_seed"]), train=(mode == "train"))
        for k, v in r.items():
            ref = z["%s/%s" % (mode, k)]
            if k in ("out/lowest_cost", "out/consistency_mask", "out/augmentation_mask"):
                assert np.array_equal(v, ref), (mode, k, float(np.mean(v != ref)))
            else:
                _close(v, ref, "%s %s" % (mode, k), rel=2e-5 if k.startswith("grad/") else 5e-6)
