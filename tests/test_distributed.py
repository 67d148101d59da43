"""world_size-2 gloo test of the sharding + all-gather path (feartracker_amd/sharding.py) on CPU.
The per-rank compute is a deterministic stand-in with the `track` API (the collective logic under test is
independent of what produces the maps); the N=1 result is the reference."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feartracker_amd.sharding import gather_maps, gather_packed, shard_range, track_local_shard, track_sharded


class FakeNet:
    """Cheap deterministic per-crop function: every crop's maps depend only on that crop."""

    def track(self, search, template_features):
        n = search.shape[0]
        s = search.reshape(n, -1)[:, :1280].reshape(n, 5, 16, 16) + template_features.reshape(n, -1)[:, :1].view(n, 1, 1, 1)
        return {"TARGET_REGRESSION_LABEL_KEY": s[:, :4].contiguous(), "TARGET_CLASSIFICATION_KEY": s[:, 4:].contiguous()}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    search = torch.randn(n, 3, 256, 256, generator=g)
    z = torch.randn(n, 256, 8, 8, generator=g)
    bbox, cls = track_sharded(FakeNet(), search, z)
    # equal-shard fast path used by bench.py
    if n % world == 0:
        lo, hi = shard_range(n, world, rank)
        out = FakeNet().track(search[lo:hi], z[lo:hi])
        g2 = gather_maps(out["TARGET_REGRESSION_LABEL_KEY"], out["TARGET_CLASSIFICATION_KEY"])
    else:
        g2 = torch.zeros(1)
    q.put((rank, bbox.numpy(), cls.numpy(), g2.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7])
def test_sharded_track_equals_single_process(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    search = torch.randn(n, 3, 256, 256, generator=g)
    z = torch.randn(n, 256, 8, 8, generator=g)
    ref = FakeNet().track(search, z)
    for rank, bbox, cls, g2 in results:
        bbox, cls, g2 = torch.from_numpy(bbox), torch.from_numpy(cls), torch.from_numpy(g2)
        assert torch.equal(bbox, ref["TARGET_REGRESSION_LABEL_KEY"])       # bit-for-bit, SURVEY.md §4
        assert torch.equal(cls, ref["TARGET_CLASSIFICATION_KEY"])
        if n % world == 0:
            assert torch.equal(g2[:, :4], ref["TARGET_REGRESSION_LABEL_KEY"])
            assert torch.equal(g2[:, 4:], ref["TARGET_CLASSIFICATION_KEY"])


def _local_shard_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n, world, rank)
    # every rank builds ONLY its own crops (the global batch never exists in one place): crop i is seeded by i
    search = torch.stack([torch.randn(3, 256, 256, generator=torch.Generator().manual_seed(i)) for i in range(lo, hi)]) \
        if hi > lo else torch.empty(0, 3, 256, 256)
    z = torch.stack([torch.randn(256, 8, 8, generator=torch.Generator().manual_seed(1000 + i)) for i in range(lo, hi)]) \
        if hi > lo else torch.empty(0, 256, 8, 8)
    bbox, cls = track_local_shard(FakeNet(), search, z, n)
    wrong = None
    try:
        track_local_shard(FakeNet(), search[:0], z[:0], n + world)          # a shard of the wrong size is refused, not padded
    except ValueError as e:
        wrong = str(e)
    q.put((rank, bbox.numpy(), cls.numpy(), wrong))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [5, 1])
def test_local_shards_never_materialise_the_global_batch(n):
    """sharding.track_local_shard: each rank hands in only its own crops (even, ragged, and an EMPTY shard on rank 1 for
    n = 1) and every rank gets the whole result, bit for bit what one process computes."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_local_shard_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    search = torch.stack([torch.randn(3, 256, 256, generator=torch.Generator().manual_seed(i)) for i in range(n)])
    z = torch.stack([torch.randn(256, 8, 8, generator=torch.Generator().manual_seed(1000 + i)) for i in range(n)])
    ref = FakeNet().track(search, z)
    for rank, bbox, cls, wrong in results:
        assert torch.equal(torch.from_numpy(bbox), ref["TARGET_REGRESSION_LABEL_KEY"])
        assert torch.equal(torch.from_numpy(cls), ref["TARGET_CLASSIFICATION_KEY"])
        assert wrong is not None and "expected" in wrong


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from feartracker_amd.sharding import OverlappedGather
    og = OverlappedGather(3, map_size=4, device="cpu")
    outs = []
    for step in range(5):                                  # 5 batches through 2 slots: every slot is reused
        buf = og.slot()
        buf.copy_(torch.full((3, 5, 4, 4), float(10 * step + rank)))
        og.launch()
        if step >= 1:                                      # result of step - 1 is complete once its slot is asked for again...
            pass
    last = og.finish().clone()                             # ...or after finish()
    # replay synchronously to get every step's expected result
    for step in range(5):
        outs.append(torch.cat([torch.full((3, 5, 4, 4), float(10 * step + r)) for r in range(world)]))
    q.put((rank, last.numpy(), outs[-1].numpy(), og.gathered[0].numpy(), outs[4].numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_double_buffering():
    """sharding.OverlappedGather (what bench.py's multi-GPU step uses): asynchronous all-gathers through two slots give the
    same rank-major result as the blocking collective, including after slot reuse."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, last, expect_last, slot0, expect_slot0 in results:
        assert (last == expect_last).all()
        assert (slot0 == expect_slot0).all()               # step 4 went through slot 0 (after steps 0 and 2)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 2048, 2049):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_rccl_gather_of_hip_maps_single_rank():
    """The real collective on the real device path: RCCL ("nccl" backend) all-gather of the maps the HIP engine wrote,
    world_size 1 (the GPU box has one GPU; world_size 2 runs on gloo above).  Exercises process-group init with a device
    id, the packed (B,5,16,16) layout and stream ordering between the engine's launches and the collective."""
    from conftest import WEIGHTS
    from feartracker_amd import FEARNetHIP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        net = FEARNetHIP(WEIGHTS, device=0, max_batch=8)
        g = torch.Generator().manual_seed(3)
        search = torch.randn(5, 3, 256, 256, generator=g).to(dev)
        z = net.get_features(torch.randn(5, 3, 128, 128, generator=g).to(dev))
        bbox, cls = net.track_maps(search, z)
        full = gather_maps(bbox, cls)
        dist.barrier()
        torch.cuda.synchronize()
        assert full.shape == (5, 5, 16, 16)
        assert torch.equal(full[:, :4], bbox) and torch.equal(full[:, 4:], cls)
        b2, c2 = track_sharded(net, search, z)
        assert torch.equal(b2, bbox) and torch.equal(c2, cls)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_ragged_shards_through_track_packed_and_rccl():
    """The N > 1 data path with the real engine on the one GPU a box has: a 7-crop global batch cut into the ragged shards of
    2, 3 and 4 ranks (shard_range), each shard written by fear_track_packed into the head of a zero-padded send buffer of the
    largest shard's size — what `track_local_shard` hands to the collective —, every send buffer through a real RCCL
    all_gather_into_tensor (1-rank group), the receive buffers trimmed and joined the way `track_local_shard` does: the result
    must be bit for bit the single-call maps, whatever the cut (so must an empty shard: 1 crop over 2 ranks)."""
    from conftest import WEIGHTS
    from feartracker_amd import FEARNetHIP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        net = FEARNetHIP(WEIGHTS, device=0, max_batch=8)
        g = torch.Generator().manual_seed(9)
        search = torch.randn(7, 3, 256, 256, generator=g).to(dev)
        z = net.get_features(torch.randn(7, 3, 128, 128, generator=g).to(dev))
        bbox, cls = net.track_maps(search, z)
        torch.cuda.synchronize()
        for n, world in ((7, 2), (7, 3), (7, 4), (1, 2)):
            cap = (n + world - 1) // world
            parts = []
            for rank in range(world):
                lo, hi = shard_range(n, world, rank)
                packed = torch.zeros((cap, 5, 16, 16), dtype=torch.float32, device=dev)
                if hi > lo:
                    net.track_packed(search[lo:hi], z[lo:hi], out=packed[: hi - lo])
                got = gather_packed(packed)                       # RCCL, this rank's slice of the receive buffer
                assert got.shape == (cap, 5, 16, 16)
                parts.append(got[: hi - lo])
                assert torch.count_nonzero(got[hi - lo:]) == 0    # the padding stays padding
            full = torch.cat(parts, dim=0)
            assert torch.equal(full[:, :4], bbox[:n]) and torch.equal(full[:, 4:], cls[:n]), (n, world)
        b2, c2 = track_local_shard(net, search, z, 7)             # world 1: the whole batch is this rank's shard
        assert torch.equal(b2, bbox) and torch.equal(c2, cls)
    finally:
        dist.destroy_process_group()


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from feartracker_amd.train_head import BoxTowerTrainHIP
    g = torch.Generator().manual_seed(100 + rank)
    grads = {"b.weight": torch.randn(7, 3, generator=g), "a.bias": torch.randn(5, generator=g), "adjust": torch.randn(1, generator=g)}
    out = BoxTowerTrainHIP.allreduce_gradients(grads)
    q.put((rank, {k: v.numpy() for k, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks():
    """N3 / BASELINE configs[4] data parallelism: the head's gradients are averaged over the ranks by ONE all-reduce of the
    flat gradient buffer (`BoxTowerTrainHIP.allreduce_gradients`; RCCL on the GPUs, gloo here) — every rank ends up with
    the mean, shapes and names preserved."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = {}
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        for k, shape in (("b.weight", (7, 3)), ("a.bias", (5,)), ("adjust", (1,))):
            want[k] = want.get(k, 0) + torch.randn(*shape, generator=g) / world
    for r in range(world):
        assert set(results[r]) == set(want)
        for k in want:
            assert results[r][k].shape == tuple(want[k].shape)
            assert torch.allclose(torch.from_numpy(results[r][k]), want[k], rtol=0, atol=1e-6)
