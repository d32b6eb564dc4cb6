#!/usr/bin/env python
"""Summarise the CONCURRENT region of a rocprofv3 kernel trace of `bench.py` with clips in flight: the last gap-delimited run of
kernels (host-side gaps > 200 us separate warm-up / single-stream / in-flight phases).  Writes a small CSV for profiles/: wall time of
the region, kernel-time sum, busy time (union of kernel intervals), average number of kernels running, hardware queues seen, and per
kernel: launches, average duration inside the region.  Usage: inflight_summary.py <dir with *_kernel_trace.csv> <out.csv> [clips]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "distribution_elementwise" in name:
        return "torch::randn/rand"
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name[:60]


def main():
    src, out = sys.argv[1], sys.argv[2]
    clips = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"))
                  for r in csv.DictReader(open(f)))
    runs, cur, cur_end = [], [rows[0]], rows[0][1]
    for r in rows[1:]:
        if r[0] - cur_end > 200_000:
            runs.append(cur)
            cur = []
        cur.append(r)
        cur_end = max(cur_end, r[1])
    runs.append(cur)
    reg = max(runs[-3:], key=len)                     # the in-flight phase is the last long run
    t0, t1 = reg[0][0], max(r[1] for r in reg)
    ksum = sum(r[1] - r[0] for r in reg)
    busy, edge = 0, t0
    for s, e, *_ in reg:
        if e > edge:
            busy += e - max(s, edge)
            edge = e
    # time with >= 2 kernels running
    ev = sorted([(s, 1) for s, *_ in reg] + [(e, -1) for _, e, *_ in reg])
    depth, last, multi = 0, t0, 0
    for t, d in ev:
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    agg = {}
    for s, e, k, q in reg:
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += e - s
    queues = sorted({q for *_, q in reg})
    wall = t1 - t0
    with open(out, "w") as o:
        o.write(f"# in-flight region of a rocprofv3 --kernel-trace run: {len(reg)} kernels on {len(queues)} hardware queues, wall {wall / 1e6:.3f} ms"
                + (f" = {wall / 1e6 / clips:.3f} ms per clip over {clips:g} clips" if clips else "") + "\n")
        o.write(f"# kernel-time sum {ksum / 1e6:.3f} ms = {ksum / wall:.2f} kernels running on average; GPU busy (union) {100 * busy / wall:.1f} % of wall; "
                f">= 2 kernels running {100 * multi / wall:.1f} % of wall\n")
        o.write("kernel,launches,avg_us_in_region,share_of_kernel_time_percent\n")
        for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"{k},{n},{d / n / 1e3:.2f},{100 * d / ksum:.2f}\n")
    print(open(out).read()[:3000])


if __name__ == "__main__":
    main()
