#!/usr/bin/env python
"""Turn the two PMC passes of scripts/pmc_traffic.sh into profiles/<name>.json: HBM bytes per launch per kernel family.
FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B?  No: rocprofv3 reports them in KB (1 unit = 1024 B is
NOT assumed here) -- we use the guide's calibration: FETCH_SIZE counts 64 B per 128-B streaming request on gfx950, so
fetched bytes = 2 * FETCH_SIZE * unit; `unit` is taken from a known-size kernel in the same run (the Whisper MLP GEMM
must read its 26.2 MB weight matrix at least once).  Usage: traffic_summary.py <dir> <out.json>"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)", name)
    return m.group(1) if m else name[:40]


def load(d, counter):
    f = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = (short(r["Kernel_Name"]), r["Grid_Size"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


def main():
    d, out = sys.argv[1], sys.argv[2]
    fetch, write = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    res = {}
    for (k, g), (n, v) in fetch.items():
        e = res.setdefault(k, {"launches": 0, "fetch_raw": 0.0, "write_raw": 0.0})
        e["launches"] += n
        e["fetch_raw"] += v
    for (k, g), (n, v) in write.items():
        res.setdefault(k, {"launches": n, "fetch_raw": 0.0, "write_raw": 0.0})["write_raw"] += v
    # calibration shape: whisper mlp1 = grid 640 blocks * 256 threads of conv_gemm_kernel
    cal = fetch.get(("conv_gemm_kernel", str(640 * 256)))
    json.dump({"by_kernel": res, "calibration_mlp1": cal, "by_grid": {f"{k}|{g}": [n, v] for (k, g), (n, v) in fetch.items()}},
              open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["fetch_raw"])[:12]:
        print(k, e)
    print("mlp1 calibration (launches, raw FETCH_SIZE sum):", cal)


if __name__ == "__main__":
    main()
