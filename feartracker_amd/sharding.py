"""Multi-GPU sharding of independent search crops.

Every search crop (with its template features) is an independent unit — `FEARNet.track` has no
cross-crop state (BN folded) — so a global batch is split contiguously over the ranks of one node
(one process per GPU, `torch.distributed` backend "nccl" = RCCL over xGMI), each rank runs the HIP
path on its shard, and ONE all-gather of the packed per-crop maps (bbox 4x16x16 + cls 1x16x16 fp32 =
5 120 B/crop) gives every rank the full result.  The reference has no inference-time collective
(single `cuda_id`, base_tracker.py:29); this module is the only place the framework communicates.

Works on CPU tensors with the gloo backend too (used by the world_size-2 tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of n crops: the first n % world ranks get one extra crop."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_maps(bbox: torch.Tensor, cls: torch.Tensor, packed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(B,4,S,S) + (B,1,S,S) -> (B,5,S,S)."""
    if packed is None:
        packed = torch.empty((bbox.shape[0], 5) + tuple(bbox.shape[2:]), dtype=bbox.dtype, device=bbox.device)
    packed[:, :4].copy_(bbox)
    packed[:, 4:].copy_(cls)
    return packed


def gather_maps(bbox: torch.Tensor, cls: torch.Tensor, packed: Optional[torch.Tensor] = None,
                gathered: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather equally sized shards: returns (world*B, 5, S, S), rank-major."""
    packed = pack_maps(bbox, cls, packed)
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype,
                               device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    return gathered


def gather_packed(packed: torch.Tensor, gathered: Optional[torch.Tensor] = None, group=None) -> torch.Tensor:
    """All-gather equally sized shards of already packed (B,5,S,S) maps (`FEARNetHIP.track_packed` writes them in that
    layout, so the collective is the only operation of the step besides the engine's own kernels)."""
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype,
                               device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    return gathered


class OverlappedGather:
    """Double-buffered all-gather of packed maps that overlaps with the NEXT batch's kernels.

    `slot()` hands out the send buffer the engine should write batch i into (`FEARNetHIP.track_packed(..., out=buf)`),
    after making the compute stream wait for the collective that last read that buffer; `launch()` starts the all-gather of
    the slot asynchronously (RCCL runs it on its own stream, ordered after the kernels enqueued so far) and returns at once, so
    the caller's next `track_packed` — into the other slot — runs while the maps travel over xGMI.  `finish()` waits for
    everything in flight and returns the most recent gathered tensor; result i is valid after `finish()` or once `slot()`
    has handed the same slot out again.  Two slots = at most one collective in flight behind the compute."""

    def __init__(self, shard_crops: int, map_size: int = 16, device=None, dtype=torch.float32, group=None, slots: int = 2):
        self.group = group
        world = dist.get_world_size(group)
        self.packed = [torch.empty((shard_crops, 5, map_size, map_size), dtype=dtype, device=device) for _ in range(slots)]
        self.gathered = [torch.empty((world * shard_crops, 5, map_size, map_size), dtype=dtype, device=device) for _ in range(slots)]
        self.work = [None] * slots
        self.i = 0
        self.last = None

    def slot(self) -> torch.Tensor:
        j = self.i % len(self.packed)
        if self.work[j] is not None:
            self.work[j].wait()            # compute stream waits for the collective that still reads packed[j]
            self.work[j] = None
        return self.packed[j]

    def launch(self) -> None:
        j = self.i % len(self.packed)
        self.work[j] = dist.all_gather_into_tensor(self.gathered[j], self.packed[j], group=self.group, async_op=True)
        self.last = j
        self.i += 1

    def finish(self) -> Optional[torch.Tensor]:
        for j, w in enumerate(self.work):
            if w is not None:
                w.wait()
                self.work[j] = None
        return None if self.last is None else self.gathered[self.last]


def track_local_shard(net, search_shard: torch.Tensor, template_shard: torch.Tensor, n_global: int, group=None):
    """Every rank hands in ONLY its own contiguous shard — crops [lo, hi) = shard_range(n_global, world, rank) of a global batch
    of n_global crops that no rank ever holds as a whole (at B = 2048 the fp32 search crops are 1.6 GB) — runs `net.track`
    on it and all-gathers the packed maps: returns (bbox (n_global,4,S,S), cls (n_global,1,S,S)) on every rank.  Ragged
    splits (n_global % world != 0) are padded to the largest shard for the collective and trimmed afterwards; a rank whose
    shard is empty still takes part in the collective."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_global, world, rank)
    if search_shard.shape[0] != hi - lo or template_shard.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} of {world} owns crops [{lo}, {hi}) of {n_global}: expected {hi - lo} crops, got "
                         f"{search_shard.shape[0]} search / {template_shard.shape[0]} template")
    from .constants import TARGET_CLASSIFICATION_KEY, TARGET_REGRESSION_LABEL_KEY
    cap = (n_global + world - 1) // world
    s_hw = int(getattr(net, "map_size", 16))                  # score-map side (16 for the 256-pixel search crop)
    dev = torch.device(getattr(net, "device", search_shard.device))
    if world > 1 and hasattr(net, "track_packed"):
        # the engine writes this rank's maps straight into the head of the (padded) send buffer
        packed = torch.zeros((cap, 5, s_hw, s_hw), dtype=torch.float32, device=dev)
        if hi > lo:
            net.track_packed(search_shard, template_shard, out=packed[: hi - lo])
    else:
        # a net without `track_packed` (any object with the reference's `track`): the shape and dtype of the send buffer come from
        # the maps themselves, and a rank with an EMPTY shard has none — so the ranks agree on them first (one tiny all-reduce:
        # map side and a dtype code, MAX over ranks, 0 from empty ones) instead of every empty rank guessing
        bbox = cls = None
        if hi > lo or world == 1:
            out = net.track(search_shard, template_shard)
            bbox, cls = out[TARGET_REGRESSION_LABEL_KEY], out[TARGET_CLASSIFICATION_KEY]
            if world == 1:
                return bbox, cls
        dtypes = [torch.float32, torch.float16, torch.bfloat16, torch.float64]
        # the device the collectives run on: CUDA whenever the group's backend can take CUDA tensors ("nccl", or a composite
        # "cuda:nccl,cpu:gloo"), for empty and non-empty ranks alike — never chosen per rank from what a rank happens to hold
        uses_cuda = "nccl" in str(dist.get_backend(group)).lower()
        meta_dev = torch.device("cuda", torch.cuda.current_device()) if uses_cuda else torch.device("cpu")
        # an unknown dtype is a code of its own (-1 sorts below every valid one, so it is reported through MIN): every rank still
        # enters the all-reduce — raising before it would leave the other ranks hanging in the collective — and all raise after it
        code_mine = (1 + dtypes.index(bbox.dtype) if bbox.dtype in dtypes else -1) if bbox is not None else 0
        meta = torch.tensor([bbox.shape[-1] if bbox is not None else 0, code_mine, -code_mine], dtype=torch.int64, device=meta_dev)
        dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
        side, code, bad = int(meta[0]), int(meta[1]), int(meta[2]) > 0
        if bad or code < 1:
            raise ValueError("a rank's score maps have an unsupported dtype (expected one of float32 / float16 / bfloat16 / float64)"
                             if bad else "no rank produced score maps")
        if bbox is not None and (bbox.shape[-1] != side or code_mine != code):
            raise ValueError("the ranks' score maps differ in size or dtype")
        if bbox is None:                                      # an empty shard still takes part in the collective
            packed = torch.zeros((cap, 5, side, side), dtype=dtypes[code - 1], device=meta_dev)
        else:
            packed = torch.zeros((cap, 5) + tuple(bbox.shape[2:]), dtype=bbox.dtype, device=meta_dev)
            pack_maps(bbox.to(meta_dev), cls.to(meta_dev), packed[: hi - lo])
    gathered = torch.empty((world * cap,) + tuple(packed.shape[1:]), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(gathered, packed, group=group)
    if n_global % world == 0:
        full = gathered
    else:
        parts = []
        for r in range(world):
            rlo, rhi = shard_range(n_global, world, r)
            parts.append(gathered[r * cap: r * cap + (rhi - rlo)])
        full = torch.cat(parts, dim=0)
    return full[:, :4].contiguous(), full[:, 4:].contiguous()


def track_sharded(net, search: torch.Tensor, template_features: torch.Tensor, group=None):
    """Convenience form for a batch that IS replicated (same tensors on every rank, any device): slices this rank's
    contiguous shard out of it and calls `track_local_shard`.  Returns (bbox (N,4,S,S), cls (N,1,S,S)) for the whole batch on
    every rank.  Production callers that shard their inputs at the source (bench.py, a data loader per rank) use
    `track_local_shard` and never materialise the global batch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = search.shape[0]
    lo, hi = shard_range(n, world, rank)
    return track_local_shard(net, search[lo:hi], template_features[lo:hi], n, group=group)
