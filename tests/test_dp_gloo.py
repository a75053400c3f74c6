"""CPU, world_size 2, gloo: the data-parallel contract of SURVEY.md 8e on the flat gradient
bucket -- identical per-rank batches reproduce the single-process gradients, distinct batches
give the mean of the per-rank gradients -- driven by the oracle loss path (the HIP path has no
CPU implementation; the bucket and the all-reduce are backend-agnostic host code)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class TinyDepthNet(torch.nn.Module):
    """stands in for the depth/pose networks: image -> (disp, axis-angle, translation)"""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(3, 1, 3, padding=1)
        self.pose = torch.nn.Linear(3, 12)

    def forward(self, img):
        disp = torch.sigmoid(self.conv(img) - 2.0)
        pose = 0.01 * self.pose(img.mean((2, 3))).view(-1, 2, 1, 6)
        return disp, pose


def _loss_and_grads(model, batch, bucket):
    from mal_amd.synthetic import to_dicts
    from oracle import mal_oracle as O
    B, _, H, W = batch["color0"].shape
    opt = O.default_opt(height=H, width=W, batch_size=B)
    inputs, mono_outputs, outputs, _ = to_dicts(batch, O.transformation_from_parameters, requires_grad=False)
    disp, pose = model(inputs[("color", 0, 0)])
    for f, i, inv in ((-1, 0, True), (1, 1, False)):
        T = O.transformation_from_parameters(pose[:, i, :, :3], pose[:, i, :, 3:], inv)
        mono_outputs[("cam_T_cam", 0, f)] = outputs[("cam_T_cam", 0, f)] = T
    mono_outputs[("disp", 0)] = disp
    outputs[("disp", 0)] = 0.5 * disp + 0.5 * outputs[("disp", 0)]
    g = torch.Generator().manual_seed(99)
    n0, n1 = torch.randn(B, 1, H, W, generator=g), torch.randn(B, 1, H, W, generator=g)
    losses, *_ = O.mal_loss_step(opt, inputs, mono_outputs, outputs, n0, n1)
    bucket.zero_()
    losses["loss"].backward()
    return float(losses["loss"])


def _worker(rank, world, port, same_batch, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mal_amd.dp import FlatGradBucket
        from mal_amd.synthetic import make_batch
        torch.manual_seed(0)
        torch.set_num_threads(2)
        model = TinyDepthNet()
        bucket = FlatGradBucket(model.parameters())
        batch = make_batch(2, 16, 24, seed=7 if same_batch else 7 + rank)
        _loss_and_grads(model, batch, bucket)
        assert bucket.check_views()
        local = bucket.flat.clone()
        bucket.all_reduce_mean()
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        if rank == 0:
            out["reduced"] = bucket.flat.numpy().copy()
            out["locals"] = [g.numpy().copy() for g in gathered]
    finally:
        dist.destroy_process_group()


def _run(same_batch):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), same_batch, out), nprocs=2, join=True)
    return dict(out)


def test_identical_batches_reproduce_single_process_gradients():
    out = _run(True)
    from mal_amd.dp import FlatGradBucket
    from mal_amd.synthetic import make_batch
    torch.manual_seed(0)
    torch.set_num_threads(2)
    model = TinyDepthNet()
    bucket = FlatGradBucket(model.parameters())
    _loss_and_grads(model, make_batch(2, 16, 24, seed=7), bucket)
    single = bucket.flat.numpy()
    assert np.abs(single).max() > 0
    assert np.allclose(out["reduced"], single, rtol=1e-6, atol=1e-9)


def test_distinct_batches_give_the_mean_of_rank_gradients():
    out = _run(False)
    mean = (out["locals"][0] + out["locals"][1]) / 2
    assert not np.allclose(out["locals"][0], out["locals"][1])
    assert np.allclose(out["reduced"], mean, rtol=1e-6, atol=1e-9)


def test_bucket_single_process_is_a_noop_and_views_hold():
    from mal_amd.dp import FlatGradBucket, shard_indices
    m = torch.nn.Linear(4, 3)
    b = FlatGradBucket(m.parameters())
    m(torch.ones(2, 4)).sum().backward()
    assert b.check_views() and b.world_size == 1 and b.all_reduce_mean() is None
    assert float(b.flat.abs().sum()) > 0
    assert list(shard_indices(10, 1, 2)) == [5, 6, 7, 8, 9]


class _Deep(torch.nn.Module):
    """a few layers, so that the flat buffer splits into pieces the backward completes one after the other; `unused`
    never receives a gradient (a frozen / unreached parameter: its piece must still be exchanged by finish())"""

    def __init__(self):
        super().__init__()
        self.a, self.b, self.c = torch.nn.Linear(6, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 3)
        self.unused = torch.nn.Linear(5, 5)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mal_amd.dp import FlatGradBucket
        res = {}
        for segs in (1, 3):
            torch.manual_seed(0)
            model = _Deep()
            bucket = FlatGradBucket(model.parameters(), segments=segs)
            x = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 + rank))
            for _ in range(2):  # two steps: the pieces re-arm
                bucket.begin_step()
                model(x).square().sum().backward()
                inside = bucket.finish()
            assert bucket.check_views()
            res[segs] = (bucket.flat.numpy().copy(), inside, len(bucket.bounds))
        if rank == 0:
            out["res"] = res
    finally:
        dist.destroy_process_group()


def test_pieces_issued_inside_backward_equal_one_all_reduce():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = out["res"]
    one, three = res[1], res[3]
    assert one[2] == 1 and one[1] == 0
    assert three[2] == 3 and three[1] >= 1        # at least one piece left from inside the backward
    assert np.array_equal(one[0], three[0])       # element-wise mean either way: bit-identical
    # the never-used parameters' gradients stay zero and were still exchanged (finish() issued their piece)
    assert float(np.abs(three[0]).sum()) > 0
