"""Multi-GPU plumbing: one process per GPU, utterances shard embarrassingly, weights are broadcast once.

The reference has no inference-time parallelism (inference is batch=1, one process; its
only distributed code is DDP training).  The path partitions into independent units (utterances /
15 s Whisper windows / 2500-frame chunks), so the only collective is the start-up broadcast of the
weight arena from rank 0 (RCCL over xGMI when the backend is "nccl"); the hot loop has none.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* (torchrun contract).
    Returns (rank, local_rank, world).  world == 1 needs no process group."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    return rank, local_rank, world


def broadcast_state_dict(sd, src=0, device="cpu", group=None):
    """Send a ``{name: tensor}`` dict from ``src`` to every rank as ONE flat fp32 arena
    (a single large broadcast is what point-to-point xGMI links like; no per-tensor collectives).
    Non-source ranks pass ``sd=None``.  Returns the dict with tensors on ``device`` (views of the arena)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return {k: v.to(device) for k, v in sd.items()}
    rank = dist.get_rank(group)
    # integer / bool buffers (e.g. BatchNorm's num_batches_tracked) are tiny and must not be rounded through fp32: they
    # travel with the layout message
    meta = [([(k, tuple(v.shape)) for k, v in sd.items() if v.is_floating_point()],
             {k: v.cpu() for k, v in sd.items() if not v.is_floating_point()})] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src, group=group)
    layout, other = meta[0]
    total = sum(int(torch.Size(s).numel()) for _, s in layout)
    if rank == src:
        arena = torch.cat([sd[k].detach().reshape(-1).to(device=device, dtype=torch.float32) for k, _ in layout])
    else:
        arena = torch.empty(total, dtype=torch.float32, device=device)
    dist.broadcast(arena, src=src, group=group)
    out, off = {}, 0
    for k, s in layout:
        n = int(torch.Size(s).numel())
        out[k] = arena[off:off + n].view(s)
        off += n
    out.update({k: v.to(device) for k, v in other.items()})
    return out


def arena_checksum(sd):
    """fp64 digest used by the tests to check every rank holds the same weights."""
    return float(sum(v.double().sum().item() + v.double().abs().sum().item() for v in sd.values()))


def shard_utterances(lengths, world):
    """Longest-processing-time assignment of utterances (cost ~ length) to ``world`` ranks.
    Returns a list of index lists, one per rank; deterministic."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    loads = [0] * world
    shards = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (loads[j], j))
        shards[r].append(i)
        loads[r] += lengths[i]
    for s in shards:
        s.sort()
    return shards


def gather_stats(value, group=None):
    """All ranks contribute one float; returns the list on every rank (end-of-run timing stats only)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [float(value)]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, float(value), group=group)
    return out
