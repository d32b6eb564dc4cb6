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
