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
        # SVCMI_DIST_BACKEND=gloo lets several ranks share ONE GPU (RCCL refuses duplicate devices): used to exercise the
        # multi-rank code path of bench.py / the folder driver on a single-GPU box; production = nccl (RCCL over xGMI)
        backend = backend or os.environ.get("SVCMI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
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


class _Ref:
    """Placeholder of one tensor inside a packed-weights skeleton: (offset, shape) into the flat fp32 arena -- or, with ``half``,
    into the fp16 side arena of ``broadcast_packed(..., fp16_filter=...)`` (offset in fp16 elements)."""
    __slots__ = ("off", "shape", "half")

    def __init__(self, off, shape, half=False):
        self.off, self.shape, self.half = off, shape, half


ARENA_ALIGN = 64        # floats: every tensor of the arena starts 256-byte aligned (the kernels need 16 bytes)


def _skeleton(obj, tensors, total, halves=None, total16=None, fp16_filter=None, name=""):
    """Copy of a container tree (dicts, lists, tuples, plain objects) with every fp32 tensor replaced by a _Ref; the tensors
    are appended to ``tensors`` in traversal order (to ``halves`` when ``fp16_filter(key, tensor)`` selects the fp16 side arena; list
    elements inherit the key of their list).
    Non-fp32 tensors and Python scalars travel inside the skeleton."""
    import copy
    rec = lambda v, key=name: _skeleton(v, tensors, total, halves, total16, fp16_filter, str(key))
    if isinstance(obj, torch.Tensor):
        if obj.dtype != torch.float32:
            return obj.detach().cpu()
        if fp16_filter is not None and fp16_filter(name, obj):
            off = (total16[0] + 2 * ARENA_ALIGN - 1) // (2 * ARENA_ALIGN) * (2 * ARENA_ALIGN)
            total16[0] = off + obj.numel()
            halves.append((off, obj))
            return _Ref(off, tuple(obj.shape), True)
        off = (total[0] + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
        total[0] = off + obj.numel()
        tensors.append((off, obj))
        return _Ref(off, tuple(obj.shape))
    if isinstance(obj, dict):
        out = obj.__class__.__new__(obj.__class__)
        dict.update(out, {k: rec(v, k) for k, v in obj.items()})
        return out
    if isinstance(obj, (list, tuple)):
        return obj.__class__(rec(v) for v in obj)
    if hasattr(obj, "__dict__") and not callable(obj):
        out = copy.copy(obj)
        out.__dict__ = {k: rec(v, k) for k, v in obj.__dict__.items()}
        return out
    return obj


def _materialise(obj, arena, wide=None):
    """``wide``: the fp16 side arena already widened to fp32 on the receiving device (one flat tensor, same offsets)."""
    if isinstance(obj, _Ref):
        n = int(torch.Size(obj.shape).numel())
        src = wide if getattr(obj, "half", False) else arena
        return src[obj.off:obj.off + n].view(obj.shape)
    if isinstance(obj, torch.Tensor):
        return obj.to(arena.device)
    if isinstance(obj, dict):
        for k in list(obj.keys()):
            dict.__setitem__(obj, k, _materialise(dict.__getitem__(obj, k), arena, wide))
        return obj
    if isinstance(obj, (list, tuple)):
        return obj.__class__(_materialise(v, arena, wide) for v in obj)
    if hasattr(obj, "__dict__") and not callable(obj):
        obj.__dict__ = {k: _materialise(v, arena, wide) for k, v in obj.__dict__.items()}
        return obj
    return obj


def pack_arena(weights, device=None, fp16_filter=None):
    """Kernel-ready weights (svcmi.weights.VitsWeights / WhisperWeights / ...: nested containers of PACKED fp32 tensors) ->
    (skeleton, flat fp32 arena).  The arena is what SURVEY.md 8e broadcasts: weight-norm already folded, GEMM layouts already
    made, so a receiving rank does no folding or packing -- it only takes views (``unpack_arena``).
    With ``fp16_filter`` the selected tensors go to a second, fp16 arena instead (returns (skeleton, arena, arena16)): the wire
    format of a model that runs those layers in fp16 anyway (``gemm_operands`` selects the large 2-D GEMM weights)."""
    tensors, total, halves, total16 = [], [0], [], [0]
    skel = _skeleton(weights, tensors, total, halves, total16, fp16_filter)
    every = tensors + halves
    dev = device if device is not None else (every[0][1].device if every else "cpu")
    arena = torch.zeros(max(total[0], 1), dtype=torch.float32, device=dev)
    for off, t in tensors:
        arena[off:off + t.numel()].copy_(t.detach().reshape(-1))
    if fp16_filter is None:
        return skel, arena
    arena16 = torch.zeros(max(total16[0], 1), dtype=torch.float16, device=dev)
    for off, t in halves:
        arena16[off:off + t.numel()].copy_(t.detach().reshape(-1))      # round-to-nearest-even, what svcmi_pack_weights_lp does for f16
    return skel, arena, arena16


def unpack_arena(skel, arena, arena16=None):
    """Rebuild the weight object around views of ``arena`` (in place in the skeleton); tensors that travelled in the fp16 side
    arena become views of ONE fp32 widening of it (their values are the fp16-rounded weights: exact for the f16 modes, which
    round the operand the same way when they pack their 16-bit image -- not for fp32 / bf16x3 use of those layers)."""
    wide = arena16.to(torch.float32) if arena16 is not None else None
    return _materialise(skel, arena, wide)


def gemm_operands(min_elements=1 << 20):
    """``fp16_filter`` for ``broadcast_packed`` -- called with (attribute / dict key of the tensor, tensor): the large 2-D GEMM
    operands, which the packed weight objects name ``*_w`` with at least ``min_elements`` elements (default 2^20: Whisper's QKV / out / MLP
    matrices, 1.6-6.6 M elements each = 99.6 % of its 1.91 GB; the conv stem's 0.3 M-element first operand stays fp32).  LayerNorm gains, biases and the positional table -- added to fp32 activations -- stay fp32."""
    return lambda name, t: name.endswith("_w") and t.dim() == 2 and t.numel() >= min_elements


def broadcast_packed(weights, src=0, device="cpu", group=None, fp16_filter=None):
    """ONE collective for a whole model: rank ``src`` passes its packed weight object, the others ``None``; everybody gets
    back an equivalent object whose tensors are views of one flat arena on ``device`` (a single large RCCL broadcast over
    xGMI; the small skeleton travels as a pickled object).  World size 1: the object is returned unchanged (with ``fp16_filter``: re-packed
    locally with the same rounding, so results do not depend on the world size).

    ``fp16_filter`` (e.g. ``gemm_operands()``; must be passed on every rank): the selected tensors travel as fp16 in a second
    message -- Whisper large-v2 0.96 GB instead of 1.91 GB ("~1.0 GB with fp16 Whisper"), for a model that is RUN in
    an f16 mode (BASELINE.json configs[4]): every rank, ``src`` included, then holds the fp16-rounded values of those operands, so all
    ranks compute identical results and the f16 kernels see exactly the operands they would have rounded themselves."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if fp16_filter is None or weights is None:
            return weights
        # the same values at every world size: a single rank also holds the fp16-rounded operands it would have received over the wire
        # (launches that stay on the fp32 kernel -- below lp_min_flops -- then compute with the same operands as in a multi-rank job)
        skel, arena, arena16 = pack_arena(weights, device, fp16_filter)
        return unpack_arena(skel, arena, arena16)
    rank = dist.get_rank(group)
    arena16 = None
    if rank == src:
        packed = pack_arena(weights, device, fp16_filter)
        skel, arena = packed[0], packed[1]
        arena16 = packed[2] if fp16_filter is not None else None
        meta = [(skel, arena.numel(), arena16.numel() if arena16 is not None else 0)]
    else:
        meta = [None]
    dist.broadcast_object_list(meta, src=src, group=group)
    skel, n, n16 = meta[0]
    if rank != src:
        arena = torch.empty(n, dtype=torch.float32, device=device)
        arena16 = torch.empty(n16, dtype=torch.float16, device=device) if n16 else None
    dist.broadcast(arena, src=src, group=group)
    if n16:
        dist.broadcast(arena16, src=src, group=group)
    return unpack_arena(skel, arena, arena16)


def plan_batches(n_utterances, world, rank, batch):
    """BASELINE.json configs[3]: ``n_utterances`` equal-length clips sharded over ``world`` ranks (LPT on equal costs = an even
    split, 512 -> 64 per GPU at 8), this rank's share cut into batches of ``batch`` (the last one may be short).
    Returns a list of utterance-id lists."""
    mine = shard_utterances([1] * n_utterances, world)[rank]
    return [mine[i:i + batch] for i in range(0, len(mine), batch)]


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
