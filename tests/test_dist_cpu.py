"""N>1 path on CPU: two gloo processes exercise the weight broadcast and the utterance sharding that
bench.py / the multi-GPU launcher use with the nccl (RCCL) backend on a node."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from svcmi import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    sd = None
    if rank == 0:
        g = torch.Generator().manual_seed(5)
        sd = {"a.weight": torch.randn(7, 5, 3, generator=g), "b.bias": torch.randn(11, generator=g), "c": torch.randn(1, 1, 12, generator=g)}
    got = D.broadcast_state_dict(sd, src=0, device="cpu")
    shards = D.shard_utterances([1000, 300, 3000, 1000, 1000, 2520, 10, 999], world)
    stats = D.gather_stats(float(rank) + 0.5)
    q.put((rank, {k: tuple(v.shape) for k, v in got.items()}, D.arena_checksum(got), shards, stats))
    torch.distributed.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, shapes0, sum0, shards0, st0), (r1, shapes1, sum1, shards1, st1) = res
    assert shapes0 == shapes1 == {"a.weight": (7, 5, 3), "b.bias": (11,), "c": (1, 1, 12)}
    assert sum0 == sum1                                   # every rank holds bit-identical weights
    assert shards0 == shards1 and st0 == st1 == [0.5, 1.5]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_lpt_sharding_covers_everything_once(world):
    lengths = [1000] * 37 + [3000] * 5 + [250, 17, 999]
    shards = D.shard_utterances(lengths, world)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(len(lengths)))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(lengths)         # LPT bound
    # 512 equal 10 s clips on 8 GPUs -> 64 each (BASELINE.json configs[3])
    assert [len(s) for s in D.shard_utterances([1000] * 512, 8)] == [64] * 8


def test_single_process_broadcast_is_identity():
    sd = {"x": torch.arange(6.0).view(2, 3)}
    out = D.broadcast_state_dict(sd, device="cpu")
    assert torch.equal(out["x"], sd["x"])


# ---------------------------------------------------------------------------------------------- folder conversion driver
class _StubConverter:
    """Stands in for the GPU models: the driver logic (sharding, outputs, stats) is what this test is about."""

    class _Hp:
        class data:
            sampling_rate = 32000

    hp = _Hp()

    def __init__(self, args, device, rank, world):
        self.rank = rank

    def convert(self, wav_path):
        import numpy as np
        n = os.path.getsize(wav_path)
        return np.full(n, float(self.rank), dtype=np.float32)


def _batch_worker(rank, world, port, folder, cwd, q):
    os.chdir(cwd)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from svcmi import svc_inference_batch as SB
    args = SB.build_parser().parse_args(["--config", "c", "--model", "m", "--wave", folder, "--spk", "s"])
    mine = SB.run_batch(args, converter_factory=_StubConverter, backend="gloo")
    q.put((rank, mine))
    torch.distributed.destroy_process_group()


def test_folder_conversion_shards_files_over_ranks(tmp_path):
    from scipy.io.wavfile import read
    folder = tmp_path / "waves"
    folder.mkdir()
    sizes = {"a.wav": 900, "b.wav": 100, "c.wav": 500, "d.wav": 450, "e.wav": 50, "notes.txt": 10}
    for name, n in sizes.items():
        (folder / name).write_bytes(b"\0" * n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, str(folder), str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res[0] + res[1]) == ["a.wav", "b.wav", "c.wav", "d.wav", "e.wav"]        # every .wav exactly once
    load = lambda fs: sum(sizes[f] for f in fs)
    assert abs(load(res[0]) - load(res[1])) <= 900                                          # LPT balance
    for r in (0, 1):
        for f in res[r]:
            sr, x = read(tmp_path / "_svc_out" / f)
            assert sr == 32000 and len(x) == sizes[f] and float(x[0]) == float(r)


# ---------------------------------------------------------------------------------------------- packed arena + configs[3] plan
def _packed_worker(rank, world, port, q):
    """BASELINE.json configs[3] on a stub: rank 0 packs the (tiny) synthesizer weights, ONE broadcast ships the arena, every
    rank rebuilds the weight object from views (no folding / packing on ranks != 0), then converts its shard of the 512
    utterances in batches of 16 with a stand-in for the GPU pipeline."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env(backend="gloo")
    from svcmi import weights as PW
    from workload import config as C, weights as W
    hp = C.tiny_hp()
    w = PW.VitsWeights(W.make_vits_state(hp, seed=1234), hp, "cpu") if rank == 0 else None
    got = D.broadcast_packed(w, src=0, device="cpu")
    assert type(got).__name__ == "VitsWeights" and got.hop == 320 and len(got.flow) == 4 and len(got.stages) == 5
    tensors = []

    def walk(o):
        if isinstance(o, torch.Tensor):
            tensors.append(o)
        elif isinstance(o, dict):
            [walk(v) for v in o.values()]
        elif isinstance(o, (list, tuple)):
            [walk(v) for v in o]
        elif hasattr(o, "__dict__"):
            [walk(v) for v in o.__dict__.values()]
    walk(got)
    assert all(t.data_ptr() % 16 == 0 for t in tensors)          # kernel operands need 16-byte alignment
    digest = float(sum(t.double().abs().sum() + 0.5 * t.double().sum() for t in tensors))
    batches = D.plan_batches(512, world, rank, 16)
    done = [i for b in batches for i in b]                        # the stub "conversion": record which utterances ran here
    q.put((rank, len(tensors), digest, [len(b) for b in batches], done))
    torch.distributed.destroy_process_group()


def test_packed_arena_broadcast_and_config3_plan_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, n0, d0, b0, done0), (_, n1, d1, b1, done1) = res
    assert n0 == n1 > 500 and d0 == d1                           # every rank holds bit-identical packed weights
    assert b0 == b1 == [16] * 16                                  # 512 utterances -> 256 per rank -> 16 batches of 16
    assert sorted(done0 + done1) == list(range(512))              # every utterance exactly once


def test_config3_plan_matches_baseline_numbers():
    for world in (1, 2, 4, 8):
        plans = [D.plan_batches(512, world, r, 16) for r in range(world)]
        assert sorted(i for p in plans for b in p for i in b) == list(range(512))
        assert all(sum(len(b) for b in p) == 512 // world for p in plans)       # 64 per GPU at 8 (BASELINE.json configs[3])
    assert [len(b) for b in D.plan_batches(40, 1, 0, 16)] == [16, 16, 8]          # a short tail batch


def test_pack_arena_round_trip_single_process():
    from svcmi import weights as PW
    from workload import config as C, weights as W
    ck = W.make_whisper_state(C.WHISPER_TINY_TEST)
    w = PW.WhisperWeights(ck, "cpu")
    skel, arena = D.pack_arena(w)
    import pickle
    w2 = D.unpack_arena(pickle.loads(pickle.dumps(skel)), arena)
    assert w2.S == w.S and w2.heads == w.heads and len(w2.blocks) == len(w.blocks)
    for a, b in zip(w.blocks, w2.blocks):
        for k in a:
            assert torch.equal(a[k], b[k]) and b[k].data_ptr() % 16 == 0
    assert torch.equal(w.pos, w2.pos) and torch.equal(w.conv2_w, w2.conv2_w)
    assert D.broadcast_packed(w) is w                             # world size 1: identity


def test_packed_arena_broadcast_and_config3_plan_world8():
    """BASELINE.json configs[3] at its real world size -- 8 ranks over gloo, ONE packed-arena broadcast, 512
    utterances -> 64 per rank -> 4 batches of 16, every utterance exactly once, bit-identical weights everywhere."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_packed_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(8))
    assert len({r[1] for r in res}) == 1 and res[0][1] > 500 and len({r[2] for r in res}) == 1      # tensor count and digest agree
    assert all(r[3] == [16] * 4 for r in res)
    assert sorted(i for r in res for i in r[4]) == list(range(512))


def _half_wire_worker(rank, world, port, q):
    """An f16 Whisper's wire format (configs[4]): large GEMM operands as fp16 in a side arena, everything else fp32."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    D.init_from_env(backend="gloo")
    from svcmi import weights as PW
    from workload import config as C, weights as W
    ck = W.make_whisper_state(C.WHISPER_TINY_TEST)
    w = PW.WhisperWeights(ck, "cpu") if rank == 0 else None
    flt = D.gemm_operands(min_elements=1 << 10)
    got = D.broadcast_packed(w, src=0, device="cpu", fp16_filter=flt)
    ref = PW.WhisperWeights(ck, "cpu")                      # (every rank can rebuild the seeded weights: the expected values)
    n_half = n_full = 0
    for a, b in zip(ref.blocks, got.blocks):
        for k in a:
            if flt(k, a[k]):
                assert torch.equal(b[k], a[k].half().float())          # the fp16-rounded operand, on EVERY rank incl. the source
                n_half += 1
            else:
                assert torch.equal(b[k], a[k])
                n_full += 1
            assert b[k].dtype == torch.float32 and b[k].data_ptr() % 16 == 0
    assert torch.equal(got.pos, ref.pos)
    q.put((rank, n_half, n_full))
    torch.distributed.destroy_process_group()


def test_fp16_wire_format_of_the_packed_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_half_wire_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:] == res[1][1:] and res[0][1] >= 4 and res[0][2] >= 4


def test_fp16_side_arena_halves_the_wire_bytes_single_process():
    from svcmi import weights as PW
    from workload import config as C, weights as W
    w = PW.WhisperWeights(W.make_whisper_state(C.WHISPER_TINY_TEST), "cpu")
    _, full = D.pack_arena(w)
    skel, arena, arena16 = D.pack_arena(w, fp16_filter=D.gemm_operands(min_elements=1 << 10))
    assert arena16.dtype == torch.float16 and arena.numel() + arena16.numel() <= full.numel() + 64 * 1024
    assert 4 * arena.numel() + 2 * arena16.numel() < 0.75 * 4 * full.numel()
    w2 = D.unpack_arena(skel, arena, arena16)
    assert torch.equal(w2.blocks[0]["qkv_w"], w.blocks[0]["qkv_w"].half().float()) and torch.equal(w2.blocks[0]["qkv_b"], w.blocks[0]["qkv_b"])
