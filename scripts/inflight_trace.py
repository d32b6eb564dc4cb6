#!/usr/bin/env python
"""What the chip does while clips are in flight: summarise a rocprofv3 --kernel-trace of
    bench.py --steps 40 --warmup 4 --no-cpu-baseline --no-roofline --no-single-stream
over a steady window of the timed region (default: the 180 ms that end 15 ms before the last kernel).
    inflight_trace.py <dir with *_kernel_trace.csv> [window_ms] [tail_ms]
Prints (a) the share of wall time with k kernels executing at once, split by whether an implicit-GEMM kernel is among them, (b) per kernel
class (name, grid) its launches per clip and mean duration IN FLIGHT -- to set beside the single-stream averages of
profiles/r*_kernel_stats.csv -- and (c) kernel-seconds per wall-second (the average number of kernels resident)."""
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
    src = sys.argv[1]
    window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
    tail_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
    f = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        wx = max(int(r.get("Workgroup_Size_X", 1) or 1), 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // wx, r.get("Queue_Id", "?")))
    mine = [r for r in rows if r[2].startswith(("conv_gemm", "attention", "snake", "splitk", "layernorm", "wn_", "coupling", "pitch", "upsample", "block_mean",
                                                 "embed", "sample_prior", "ncl_to", "torch::randn"))]
    t_last = max(r[1] for r in mine)
    t1 = t_last - int(tail_ms * 1e6)
    t0 = t1 - int(window_ms * 1e6)
    win = [r for r in mine if r[1] > t0 and r[0] < t1]
    queues = sorted({r[4] for r in win})
    # clips in the window: one pitch_prefix_kernel launch per clip
    clips = sum(1 for r in win if r[2].startswith("pitch_prefix")) or 1
    print(f"window {window_ms:.0f} ms ending {tail_ms:.0f} ms before the last kernel: {len(win)} kernel executions on queues {queues}, {clips} clips "
          f"=> {window_ms / clips:.3f} ms per clip")
    # (a) sweep line
    ev = []
    for (s, e, k, g, q) in win:
        s, e = max(s, t0), min(e, t1)
        gemm = k.startswith("conv_gemm")
        ev.append((s, 1, gemm))
        ev.append((e, -1, gemm))
    ev.sort(key=lambda x: (x[0], x[1]))
    hist, hist_g, run, run_g, prev = {}, {}, 0, 0, t0
    for (t, d, gemm) in ev:
        dt = t - prev
        if dt > 0:
            hist[run] = hist.get(run, 0) + dt
            hist_g[run_g] = hist_g.get(run_g, 0) + dt
            if run and not run_g:
                hist["nogemm"] = hist.get("nogemm", 0) + dt
        run += d
        run_g += d if gemm else 0
        prev = t
    tot = float(t1 - t0)
    print("kernels executing at once : " + "  ".join(f"{k}: {hist.get(k, 0) / tot:.3f}" for k in range(0, 7)) +
          f"  (>=7: {sum(v for k, v in hist.items() if isinstance(k, int) and k >= 7) / tot:.3f})")
    print("implicit-GEMM kernels at once: " + "  ".join(f"{k}: {hist_g.get(k, 0) / tot:.3f}" for k in range(0, 5)))
    print(f"wall time with kernels running but NO implicit GEMM among them: {hist.get('nogemm', 0) / tot:.3f}")
    busy = sum(min(e, t1) - max(s, t0) for (s, e, k, g, q) in win)
    print(f"kernel-seconds per wall-second (mean kernels resident): {busy / tot:.2f}")
    # (b) per class
    agg = {}
    for (s, e, k, g, q) in win:
        if s < t0 or e > t1:
            continue
        a = agg.setdefault((k, g), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print("kernel,grid,launches_per_clip,mean_us_in_flight,kernel_ms_per_clip")
    for (k, g), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{k},{g},{n / clips:.1f},{us / n:.2f},{us / clips / 1e3:.3f}")
    fam = sum(us for (k, g), (n, us) in agg.items() if k.startswith("conv_gemm")) / clips / 1e3
    allk = sum(us for (k, g), (n, us) in agg.items()) / clips / 1e3
    print(f"implicit-GEMM family: {fam:.3f} kernel-ms per clip in flight; all kernels {allk:.3f} kernel-ms per clip")


if __name__ == "__main__":
    main()
