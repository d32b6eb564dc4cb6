#!/usr/bin/env python
"""The in-flight timeline of the implicit-GEMM family, reconstructed from the records the kernels write themselves.

rocprofv3 serialises the four HIP queues of the judged launch regime (4 clips in flight), so the regime the judged line is measured in had
no kernel-level evidence.  A probe build of the library (scripts/build_variant.sh timeline -DSVCMI_PROBE_KTRACE=1; never the shipped one)
makes EVERY block of every implicit-GEMM launch append {s_memrealtime at entry, at the end of its epilogue, output pointer, n_out << 32 | K,
grid << 32 | rows} to a device ring (csrc/conv_gemm_body.h); `bench.py` with SVCMI_TIMELINE=<out.json> and SVCMI_LIB=<that build> runs its
timed loop, then records a few more steps and hands the records to `summarise` below.

    launches      blocks with the same output pointer whose entries lie within `gap_us` of each other = one launch: [first entry, last exit]
    busy          union of the launch intervals = wall time with at least one GEMM launch resident on the chip
    gemm_ms_per_step = busy / clips completed in the window (<= ms_per_step by construction); sum_ms_per_step = sum of launch durations / clips
    concurrency   share of the window with k GEMM launches resident
    classes       per (n_out, K, rows, grid): launches per clip, mean duration in flight (to set beside the one-clip-at-a-time record)
Ticks are s_memrealtime's: 100 MHz (10 ns)."""
import json
import sys

import numpy as np

TICK_US = 0.01


def launches_from_records(rec, gap_us=2000.0):
    """rec: uint64 [n, 5] block records -> list of dicts(start, end, key, n_out, k, rows, grid, blocks), sorted by start.
    Blocks of one launch share (output pointer, shape, grid); launches with the same output pointer (the same lane's workspace slot, reused
    by every layer) never overlap, so within a key the blocks sorted by entry time are cut every `grid` records; a group that meets a gap of
    more than `gap_us` before it is complete was cut by the start / end of the recording and keeps its short block count."""
    rec = np.asarray(rec, dtype=np.uint64).reshape(-1, 5)
    out = []
    if rec.shape[0] == 0:
        return out
    order = np.lexsort((rec[:, 0], rec[:, 4], rec[:, 3], rec[:, 2]))
    rec = rec[order]
    gap = int(gap_us / TICK_US)
    n = rec.shape[0]

    def close(a, b):
        blk = rec[a:b]
        out.append(dict(start=int(blk[:, 0].min()), end=int(blk[:, 1].max()), key=int(blk[0, 2]), n_out=int(blk[0, 3] >> np.uint64(32)),
                        k=int(blk[0, 3] & np.uint64(0xffffffff)), rows=int(blk[0, 4] & np.uint64(0xffffffff)), grid=int(blk[0, 4] >> np.uint64(32)),
                        blocks=int(b - a)))

    start = 0
    for i in range(1, n + 1):
        grid = int(rec[start, 4] >> np.uint64(32))
        new_key = i == n or rec[i, 2] != rec[start, 2] or rec[i, 3] != rec[start, 3] or rec[i, 4] != rec[start, 4]
        if new_key or i - start == grid or int(rec[i, 0]) - int(rec[i - 1, 0]) > gap:
            close(start, i)
            start = i
    out.sort(key=lambda d: d["start"])
    return out


def union_ticks(intervals):
    total, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                total += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        total += cur_e - cur_s
    return total


def concurrency_shares(intervals, t0, t1, kmax=4):
    """share of [t0, t1] with k intervals open (k = 0 .. kmax, last bucket = kmax or more)"""
    ev = []
    for s, e in intervals:
        s, e = max(s, t0), min(e, t1)
        if e > s:
            ev.append((s, 1))
            ev.append((e, -1))
    ev.sort()
    shares = [0] * (kmax + 1)
    k, last = 0, t0
    for t, d in ev:
        shares[min(k, kmax)] += t - last
        last = t
        k += d
    shares[min(k, kmax)] += t1 - last
    tot = float(max(t1 - t0, 1))
    return [x / tot for x in shares]


def summarise(rec, clips, marker=None, gap_us=2000.0, peak_tflops=157.3, complete_only=True):
    """rec: block records of a window in which `clips` clips were converted (steps x batch; all lanes together).  `marker` = (n_out, K): launches
    of that class happen once per clip (e.g. the first Whisper convolution) and are used to trim the window to whole clips when given."""
    ls = launches_from_records(rec, gap_us)
    if complete_only:
        ls = [l for l in ls if l["blocks"] == l["grid"]]           # (launches cut by the start / end of the recording or the ring's capacity)
    if not ls:
        return dict(launches=0)
    t0, t1 = min(l["start"] for l in ls), max(l["end"] for l in ls)
    iv = [(l["start"], l["end"]) for l in ls]
    busy = union_ticks(iv)
    dur = sum(e - s for s, e in iv)
    flops = sum(2.0 * l["rows"] * l["n_out"] * l["k"] for l in ls)
    classes = {}
    for l in ls:
        c = classes.setdefault((l["n_out"], l["k"], l["rows"], l["grid"]), [0, 0])
        c[0] += 1
        c[1] += l["end"] - l["start"]
    cls = [dict(n_out=k[0], k=k[1], rows=k[2], grid=k[3], launches_per_clip=round(v[0] / clips, 3), mean_us=round(v[1] / v[0] * TICK_US, 2),
                ms_per_clip=round(v[1] * TICK_US / 1e3 / clips, 4)) for k, v in classes.items()]
    cls.sort(key=lambda d: -d["ms_per_clip"])
    window_ms = (t1 - t0) * TICK_US / 1e3
    return dict(launches=len(ls), clips=clips, window_ms=round(window_ms, 3), ms_per_step_in_window=round(window_ms / clips, 4),
                gemm_ms_per_step=round(busy * TICK_US / 1e3 / clips, 4), sum_ms_per_step=round(dur * TICK_US / 1e3 / clips, 4),
                gemm_gflop_per_step=round(flops / clips / 1e9, 1),
                frac_while_resident=round(flops / (busy * TICK_US * 1e-6) / 1e12 / peak_tflops, 4),
                frac_sum_of_launches=round(flops / (dur * TICK_US * 1e-6) / 1e12 / peak_tflops, 4),
                concurrency_share=[round(x, 4) for x in concurrency_shares(iv, t0, t1)], classes=cls[:24])


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    clips = int(sys.argv[2])
    print(json.dumps(summarise(raw, clips), indent=1))


if __name__ == "__main__":
    main()
