"""GPU: the trainer-compatible harness (mal_amd.harness) -- RepDepth forward with the HIP cost volume, the loss
path in one call, backward to the parameters through the flat gradient bucket, Adam, checkpoints."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="session", autouse=True)
def _built():
    from mal_amd import build
    build.build(verbose=False)


def test_train_steps_and_checkpoint(tmp_path):
    import random
    from mal_amd import harness
    torch.manual_seed(0)
    random.seed(0)
    opt = harness.default_options(batch_size=2, height=96, width=160)
    h = harness.TrainHarness(opt, DEV)
    inputs = harness.synthetic_inputs(opt, DEV, seed=3)
    before = [p.detach().clone() for p in h.params[:3]]
    losses = [float(h.train_step(inputs)["loss"].detach()) for _ in range(3)]
    assert all(torch.isfinite(torch.tensor(losses)))
    assert h.bucket.check_views()                       # gradients still live in the one flat buffer
    assert float(h.bucket.flat.abs().sum()) > 0          # ... and reached the parameters
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, h.params[:3]))
    lo, hi = h.tracker.compute()
    assert 0.1 <= float(lo) < float(hi)
    h.save(str(tmp_path))
    for f in ("model.pth", "track.pth", "adam.pth"):
        assert (tmp_path / f).is_file()
    h2 = harness.TrainHarness(opt, DEV)
    h2.load(str(tmp_path))
    for a, b in zip(h.model.state_dict().values(), h2.model.state_dict().values()):
        assert torch.equal(a, b)
    assert torch.equal(h2.tracker.compute()[1].cpu(), hi.cpu())


def test_process_batch_keys_match_upstream():
    import random
    from mal_amd import harness
    random.seed(1)
    opt = harness.default_options(batch_size=2, height=96, width=160)
    h = harness.TrainHarness(opt, DEV)
    outputs, losses = h.process_batch(harness.synthetic_inputs(opt, DEV, seed=4))
    for k in (("disp", 0), ("mono_disp", 0), ("axisangle", 0, -1), ("translation", 0, 1), ("cam_T_cam", 0, -1),
              "lowest_cost", "consistency_mask", "augmentation_mask"):
        assert k in outputs, k
    assert outputs["lowest_cost"].shape == (2, 96, 160) and outputs[("disp", 0)].shape == (2, 1, 96, 160)
    for k in ("loss", "reproj_loss/0", "consistency_loss/0", "distil_loss", "mono/loss"):
        assert k in losses and torch.isfinite(losses[k]).all()


def test_training_reduces_the_loss_on_a_fixed_batch():
    """end to end: networks -> HIP loss step -> backward through the flat bucket -> Adam; overfitting one batch
    must drive the total loss down (gradients of the HIP path are descent directions for the networks)"""
    import random
    from mal_amd import harness
    torch.manual_seed(1)
    random.seed(1)
    opt = harness.default_options(batch_size=2, height=96, width=160, no_matching_augmentation=True, learning_rate=2e-4)
    h = harness.TrainHarness(opt, DEV)
    inputs = harness.synthetic_inputs(opt, DEV, seed=5)
    losses = [float(h.train_step(inputs)["loss"].detach()) for _ in range(25)]
    head, tail = sum(losses[:3]) / 3, sum(losses[-3:]) / 3
    assert tail < 0.95 * head, losses
