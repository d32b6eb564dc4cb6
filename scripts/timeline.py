#!/usr/bin/env python
"""Timeline of ONE graph replay from a rocprofv3 kernel trace: finds the last step (gap-delimited), prints per-segment
wall spans (Whisper / prior encoder + flow / generator stages) and, with -v, every kernel with start offset and stream.
Usage: timeline.py <dir with *_kernel_trace.csv> [-v]"""
import csv
import glob
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name[:50]


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
             int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Queue_Id"]) for r in csv.DictReader(open(f))]
    rows.sort()
    # steps are separated by host-side gaps > 200 us; take the last run of >= 300 kernels
    runs, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - max(r[1] for r in cur[-8:]) > 200_000:
            runs.append(cur)
            cur = []
        cur.append(b)
    runs.append(cur)
    runs = [r for r in runs if len(r) >= 300]
    step = runs[-1] if len(sys.argv) < 4 else runs[int(sys.argv[3])]
    t0 = step[0][0]
    end = max(r[1] for r in step)
    print(f"{len(runs)} steps found; last one: {len(step)} kernels, wall {(end - t0) / 1e3:.1f} us, "
          f"kernel-time sum {sum(r[1] - r[0] for r in step) / 1e3:.1f} us")
    # busy time (union of intervals) and gaps
    busy, cur_e = 0, t0
    for s, e, *_ in step:
        if e > cur_e:
            busy += e - max(s, cur_e)
            cur_e = e
    print(f"union-busy {busy / 1e3:.1f} us, idle gaps {(end - t0 - busy) / 1e3:.1f} us")
    if "-v" in sys.argv:
        for s, e, n, g, q in step:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q:>3s} {n:45s} grid {g}")


if __name__ == "__main__":
    main()
