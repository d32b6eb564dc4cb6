#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (kernel trace) into a small CSV: per-kernel calls / total / average,
plus a per-(kernel, grid) breakdown for the svcmi kernels.  Usage: rocpd_summary.py <results.db> <out.csv> [steps]"""
import sqlite3
import sys


def short(name):
    for key, s in (("conv_gemm_kernel", "conv_gemm_kernel"), ("splitk_reduce", "splitk_reduce_kernel"),
                   ("attention_kernel", "attention_kernel"), ("snake_alias", "snake_alias_kernel"),
                   ("layernorm", "layernorm_kernel"), ("distribution", "torch::randn/rand"), ("copyBuffer", "rocclr_copyBuffer")):
        if key in name:
            if "<" in name and key in ("conv_gemm_kernel", "attention_kernel"):
                return s + "<" + name.split("<")[1].split(">")[0].replace(" ", "") + ">"
            return s
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:60]


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x/workgroup_x, grid_y, grid_z, duration from kernels").fetchall()
    agg, by_grid = {}, {}
    for name, gx, gy, gz, dur in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += dur
        if any(t in k for t in ("conv_gemm", "attention", "snake")):
            g = by_grid.setdefault((k, gx, gy, gz), [0, 0.0])
            g[0] += 1
            g[1] += dur
    total = sum(a[1] for a in agg.values())
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in us; per_step = calls / %g traced steps)\n" % steps)
        f.write("kernel,calls,calls_per_step,total_us,avg_us,percent\n")
        for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k},{n},{n / steps:.1f},{d / 1e3:.1f},{d / n / 1e3:.2f},{100 * d / total:.2f}\n")
        f.write("\n# per (kernel, grid) -- grid in workgroups\nkernel,grid_x,grid_y,grid_z,calls_per_step,avg_us,ms_per_step\n")
        for (k, gx, gy, gz), (n, d) in sorted(by_grid.items(), key=lambda kv: -kv[1][1])[:40]:
            f.write(f"{k},{gx},{gy},{gz},{n / steps:.1f},{d / n / 1e3:.2f},{d / steps / 1e6:.3f}\n")
    print(open(out).read()[:1800])


if __name__ == "__main__":
    main()
