"""bench.py's launch contract (host logic, no GPU): `--gpus N` must never print a line for a job of another size."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)


def test_gpus_n_without_launcher_and_without_n_gpus_exits_nonzero_and_prints_no_line():
    """`python bench.py --gpus 2` with no WORLD_SIZE: it would re-exec itself under torch.distributed.run with 2 ranks; on a box with
    fewer than 2 GPUs (this container: none) it must refuse instead of measuring one GPU under an N = 2 flag."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("box has >= 2 GPUs: the re-exec path would run the real benchmark")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert r.stdout.strip() == ""
    assert "not measuring a smaller job" in r.stderr


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout.strip() == ""
    assert "WORLD_SIZE 1 != --gpus 4" in r.stderr


def test_the_drivers_eight_rank_command_runs_end_to_end_on_cpu():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8
    --config 3 ...` -- the command the driver launches on an 8-GPU node -- with `--dry-run`: gloo on CPU, tiny weights, a stand-in pipeline, but the
    REAL rendezvous, packed-arena broadcasts, configs[3] sharding (512 utterances -> 64 per rank -> 4 batches of 16), barriers around the timed
    region, max-over-ranks timing and the all-reduce that counts the ranks.  Rank 0 prints exactly one JSON line."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3", "--steps", "2", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-400:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["dry_run"] is True and d["value"] is None and d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 2 and d["warmup"] == 1
    assert c["rccl_ranks_seen"] == 8 and c["dist_backend"] == "gloo" and c["utterances_per_rank"] == [64] * 8 and c["batches_on_rank0"] == [16] * 4
    assert c["weights_identical_on_every_rank"] is True and c["every_utterance_exactly_once"] is True


def test_dry_run_self_spawns_without_a_launcher():
    """`python bench.py --gpus 2 --dry-run` with no WORLD_SIZE re-executes itself as 2 ranks (gloo) and prints one line for the 2-rank job."""
    import json
    r = _run(["--gpus", "2", "--config", "3", "--steps", "1", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stderr[-600:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks_seen"] == 2 and d["config"]["utterances_per_rank"] == [256, 256]
