#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace --stats --output-format csv` run into one small CSV for profiles/:
per-kernel calls / total / average (names shortened), then a per-(kernel, grid, block) breakdown with register
counts.  Usage: prof_summary.py <dir with *_kernel_trace.csv> <out.csv> [traced_steps]"""
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workload.stamp import csrc_sha  # noqa: E402


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if "distribution_elementwise" in name:
        return "torch::randn/rand"
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>(]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name[:60]


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    agg, by_grid = {}, {}
    rows = list(csv.DictReader(open(f)))
    # model load (H->D weight upload = __amd_rocclr_copyBuffer bursts, fills) ends where the first svcmi kernel starts; kernels that
    # only ever ran before that point are one-time set-up, not part of a step
    first_step = min((int(r["Start_Timestamp"]) for r in rows if "anonymous namespace" in r["Kernel_Name"]), default=0)
    for r in rows:
        k = short(r["Kernel_Name"])
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(k, [0, 0.0, True])
        a[0] += 1
        a[1] += dur
        a[2] = a[2] and int(r["End_Timestamp"]) <= first_step
        wx = max(int(r["Workgroup_Size_X"]), 1)
        key = (k, int(r["Grid_Size_X"]) // wx, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]), wx,
               r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
        g = by_grid.setdefault(key, [0, 0.0])
        g[0] += 1
        g[1] += dur
    total = sum(a[1] for k, a in agg.items() if "spin_kernel" not in k and not a[2])
    with open(out, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary; durations in us; {steps:g} traced steps; "
                f"total kernel time per step {total / steps / 1e3:.3f} ms (spin_kernel and one-time model-load dispatches excluded); csrc_sha {csrc_sha()}\n")
        o.write("kernel,calls,calls_per_step,total_us,avg_us,percent\n")
        for k, (n, d, setup) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if setup:
                o.write(f"{k} [model load: all {n} dispatches precede the first step],{n},0.0,{d:.1f},{d / n:.2f},0.00\n")
            else:
                o.write(f"{k},{n},{n / steps:.1f},{d:.1f},{d / n:.2f},{100 * d / total:.2f}\n")
        o.write("\n# per (kernel, grid in workgroups, block) -- top 50 by time\n")
        o.write("kernel,grid_x,grid_y,grid_z,block,vgpr,agpr,sgpr,lds,calls_per_step,avg_us,ms_per_step\n")
        setup = {k for k, a in agg.items() if a[2]} | {k for k in agg if "spin_kernel" in k}      # one-time dispatches are not per-step rows
        for key, (n, d) in sorted(((k, v) for k, v in by_grid.items() if k[0] not in setup), key=lambda kv: -kv[1][1])[:50]:
            o.write(",".join(str(x) for x in key) + f",{n / steps:.1f},{d / n:.2f},{d / steps / 1e3:.3f}\n")
    print(open(out).read()[:6000])


if __name__ == "__main__":
    main()
