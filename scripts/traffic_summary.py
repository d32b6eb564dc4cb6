#!/usr/bin/env python
"""Turn the two PMC passes of scripts/pmc_traffic.sh into profiles/<name>_traffic.json: HBM bytes per launch per kernel.
Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are reported in KB; on gfx950 the 128-byte
streaming requests are tallied at 64 B, so fetched bytes = 2 x FETCH_SIZE x 1024 (sanity check printed below: the Whisper
MLP up-projection must read its 26.2 MB weight matrix at least once); WRITE_SIZE is taken as is (x 1024).
`conv_gemm_kernel` in the output is the union of the single and the grouped launches of the same kernel body (what
bench.py's roofline object refers to).  Usage: traffic_summary.py <dir> <out.json> <traced_steps>"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from workload.stamp import csrc_sha  # noqa: E402


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)", name)
    return m.group(1) if m else name[:40]


def load(d, counter):
    f = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    agg, by_grid = {}, {}
    seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, [0, 0.0])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:          # one row per (dispatch, XCD/instance): count the dispatch once, sum the values
            seen.add(key)
            a[0] += 1
        a[1] += float(r["Counter_Value"])
        g = by_grid.setdefault((k, r["Grid_Size"]), [set(), 0.0])
        g[0].add(r.get("Dispatch_Id"))
        g[1] += float(r["Counter_Value"])
    return agg, by_grid


def main():
    d, out, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
    (fetch, fgrid), (write, _) = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        n = fetch.get(k, write.get(k))[0]
        rd = 2.0 * 1024.0 * fetch.get(k, [0, 0.0])[1]
        wr = 1024.0 * write.get(k, [0, 0.0])[1]
        kernels[k] = {"launches_per_step": round(n / steps, 1), "hbm_read_bytes_per_launch": int(rd / max(n, 1)),
                      "hbm_write_bytes_per_launch": int(wr / max(n, 1)), "hbm_read_GB_per_step": round(rd / steps / 1e9, 3),
                      "hbm_write_GB_per_step": round(wr / steps / 1e9, 3)}
    fam = [k for k in kernels if k.startswith("conv_gemm")]
    if len(fam) > 1 or (fam and fam[0] != "conv_gemm_kernel"):
        n = sum(kernels[k]["launches_per_step"] for k in fam)
        rd = sum(kernels[k]["hbm_read_GB_per_step"] for k in fam) * 1e9
        wr = sum(kernels[k]["hbm_write_GB_per_step"] for k in fam) * 1e9
        for k in fam:
            kernels[k + " (part)"] = kernels.pop(k)
        kernels["conv_gemm_kernel"] = {"launches_per_step": round(n, 1), "hbm_read_bytes_per_launch": int(rd / n),
                                       "hbm_write_bytes_per_launch": int(wr / n), "hbm_read_GB_per_step": round(rd / 1e9, 3),
                                       "hbm_write_GB_per_step": round(wr / 1e9, 3), "note": "single + grouped launches"}
    total_r = sum(v["hbm_read_GB_per_step"] for k, v in kernels.items() if "(part)" not in k)
    total_w = sum(v["hbm_write_GB_per_step"] for k, v in kernels.items() if "(part)" not in k)
    json.dump({"csrc_sha": csrc_sha(), "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over "
                         f"`bench.py --steps 2 --warmup 1 --eager`; {steps:g} traced steps (scripts/pmc_traffic.sh)",
               "units": "KB counters; fetched bytes = 2 x FETCH_SIZE x 1024 (gfx950: 128-B streaming requests tallied at 64 B), "
                        "written bytes = WRITE_SIZE x 1024",
               "traced_steps": steps, "hbm_read_GB_per_step": round(total_r, 3), "hbm_write_GB_per_step": round(total_w, 3),
               "kernels": kernels}, open(out, "w"), indent=1)
    for k, e in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_read_GB_per_step"])[:14]:
        print(f"{k:44s} {e}")
    print(f"step total: read {total_r:.2f} GB, written {total_w:.2f} GB")
    mlp = [(g, len(ids), v) for (k, g), (ids, v) in fgrid.items() if k.startswith("conv_gemm_kernel") and g == str(512 * 256)]
    for g, n, v in mlp:
        print(f"check: conv_gemm_kernel grid {g}: {n} launches, {2 * 1024 * v / max(n, 1) / 1e6:.1f} MB fetched per launch "
              f"(Whisper MLP: 26.2 MB of weights each)")


if __name__ == "__main__":
    main()
