"""scripts/inflight_timeline.py on synthetic block records: two lanes whose GEMM launches overlap by a known amount (no GPU needed)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import inflight_timeline as TL  # noqa: E402


def _launch(t0_us, dur_us, key, n_out, k, rows, grid, rng):
    """grid blocks that start within 2 us of t0 and end within 3 us before t0 + dur (the last one exactly there)"""
    s = (t0_us + rng.uniform(0, 2.0, grid)) / TL.TICK_US
    e = (t0_us + dur_us - rng.uniform(0, 3.0, grid)) / TL.TICK_US
    s[0], e[-1] = t0_us / TL.TICK_US, (t0_us + dur_us) / TL.TICK_US
    r = np.zeros((grid, 5), dtype=np.uint64)
    r[:, 0], r[:, 1], r[:, 2] = s.astype(np.uint64), e.astype(np.uint64), key
    r[:, 3] = (np.uint64(n_out) << np.uint64(32)) | np.uint64(k)
    r[:, 4] = (np.uint64(grid) << np.uint64(32)) | np.uint64(rows)
    return r


def test_two_lanes_with_known_overlap():
    rng = np.random.default_rng(0)
    recs = []
    # lane A: per clip one 60 us launch (n_out 5120) at t and one 20 us launch (n_out 1280) at t + 100; lane B the same, shifted by 30 us
    for clip in range(3):
        for lane, shift in ((0, 0.0), (1, 30.0)):
            t = 1000.0 + 2000.0 * clip + shift
            recs.append(_launch(t, 60.0, 0x1000 + lane, 5120, 1280, 500, 512, rng))
            recs.append(_launch(t + 100.0, 20.0, 0x2000 + lane, 1280, 1280, 500, 256, rng))
    rec = np.concatenate(recs)
    rng.shuffle(rec)                                        # blocks arrive in any order
    ls = TL.launches_from_records(rec)
    assert len(ls) == 12 and all(l["blocks"] == l["grid"] for l in ls)
    s = TL.summarise(rec, clips=6)
    # per pair of clips: the 60 us launches overlap by 30 us (union 90), the 20 us ones are disjoint from everything (t + 100 .. 120, t + 130 .. 150)
    assert abs(s["gemm_ms_per_step"] - (90.0 + 40.0) / 2 / 1e3) < 2e-4
    assert abs(s["sum_ms_per_step"] - 80.0 / 1e3) < 2e-4
    assert s["gemm_ms_per_step"] <= s["ms_per_step_in_window"]
    flops = 2.0 * 500 * 5120 * 1280 + 2.0 * 500 * 1280 * 1280
    assert abs(s["gemm_gflop_per_step"] - flops / 1e9) < 0.1
    assert abs(s["frac_sum_of_launches"] - flops / 80e-6 / 1e12 / 157.3) < 2e-3
    shares = s["concurrency_share"]
    assert abs(sum(shares) - 1.0) < 1e-3 and shares[2] > 0 and shares[3] == 0
    top = s["classes"][0]
    assert (top["n_out"], top["launches_per_clip"]) == (5120, 1.0) and abs(top["mean_us"] - 60.0) < 0.1


def test_cut_launches_are_dropped():
    rng = np.random.default_rng(1)
    full = _launch(100.0, 50.0, 7, 80, 240, 20000, 313, rng)
    cut = _launch(400.0, 50.0, 8, 80, 240, 20000, 313, rng)[:100]           # the recording ended inside this launch
    s = TL.summarise(np.concatenate([full, cut]), clips=1)
    assert s["launches"] == 1


def test_same_workspace_slot_reused_by_consecutive_layers():
    """One lane, the same output pointer for every layer (the arena hands the same slot out again), launches 80 us apart: cut by the grid count."""
    rng = np.random.default_rng(2)
    rec = np.concatenate([_launch(100.0 + 80.0 * i, 60.0, 0x99, 3840, 1280, 500, 480, rng) for i in range(5)])
    ls = TL.launches_from_records(rec)
    assert len(ls) == 5 and all(l["blocks"] == 480 for l in ls)
    assert all(abs((l["end"] - l["start"]) * TL.TICK_US - 60.0) < 0.05 for l in ls)
