#!/usr/bin/env python
"""Summarise scripts/pmc_bench.sh into profiles/<name>.json: per kernel family MFMA-pipe utilisation, wave-state split and
LDS bank-conflict share.  Units (MI355X_MICROARCH.md): SQ_VALU_MFMA_BUSY_CYCLES = cycles summed over the 1024 SIMDs;
GRBM_GUI_ACTIVE = cycles summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* = quad-cycles per wave."""
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
    m = re.match(r"([A-Za-z0-9_:]+)(<[^>(]*>)?", name)      # keep the template arguments: tile policies are separate rows
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else name[:40]


def load(d):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg.setdefault(short(r["Kernel_Name"]), {})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp = a.setdefault("_dispatches", set())
            disp.add((f, r["Dispatch_Id"]))
    return agg


def main():
    d, out = sys.argv[1], sys.argv[2]
    agg = load(d)
    res = {}
    for k, c in agg.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 2.0        # collected in both passes
        if gui <= 0:
            continue
        simd_cycles = gui / 8.0 * 1024.0
        wave = c.get("SQ_WAVE_CYCLES", 0.0)
        e = {"dispatches_per_pass": len(c["_dispatches"]) // 2,
             "mfma_pipe_utilisation": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4),
             "busy_cycles_per_xcd": round(gui / 8.0)}
        if wave > 0:
            e["wave_cycles_share"] = {"issue_stall(SQ_WAIT_INST_ANY)": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 3),
                                      "parked(SQ_WAIT_ANY)": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 3),
                                      "issuing(SQ_ACTIVE_INST_ANY)": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 3)}
        if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0:
            e["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        res[k] = e
    json.dump({"csrc_sha": csrc_sha(), "source": "rocprofv3 --pmc (two passes, --kernel-trace only) over `bench.py --steps 1 --warmup 1 --eager`",
               "note": "mfma_pipe_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): cycle-based, i.e. "
                       "relative to the clock the chip actually ran at",
               "kernels": res}, open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["busy_cycles_per_xcd"])[:10]:
        print(k, json.dumps(e))


if __name__ == "__main__":
    main()
