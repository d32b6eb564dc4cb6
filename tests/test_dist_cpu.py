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
