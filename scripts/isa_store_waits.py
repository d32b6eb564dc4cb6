#!/usr/bin/env python
"""Which loops of which kernels wait for their own stores?  gfx9 counts vector loads and stores in ONE counter (vmcnt), so a loop that
loads, computes and stores, and whose loads sit behind a branch, gets `s_waitcnt vmcnt(0)` at its head from hipcc -- every iteration then
waits for the previous iteration's store to be acknowledged by the memory system (round 6: 3600 cycles per iteration in the GEMM epilogue).
    python scripts/isa_store_waits.py file.s [...]      (hipcc --cuda-device-only -S output)
Lists, per kernel, every innermost loop that contains a store and a vmcnt(0) wait."""
import re
import sys


def kernels(text):
    src = text.split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    for s in starts:
        e = next(i for i in range(s + 1, len(src)) if src[i].startswith(".Lfunc_end"))
        yield src[s][:-1], [l.split(";")[0].rstrip() for l in src[s:e] if l.split(";")[0].strip()]


def loops(body):
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    out = []
    for k, l in enumerate(body):
        m = re.search(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= k:
            out.append((labels[m.group(1)], k))
    return [(a, b) for (a, b) in out if not any(a <= c and d <= b and (c, d) != (a, b) for (c, d) in out)]      # innermost only


def main():
    for path in sys.argv[1:]:
        for name, body in kernels(open(path).read()):
            for a, b in loops(body):
                seg = body[a:b + 1]
                stores = sum(bool(re.search(r"(global|buffer|flat)_store", l)) for l in seg)
                loads = sum(bool(re.search(r"(global|buffer|flat)_load", l)) for l in seg)
                w0 = sum("vmcnt(0)" in l for l in seg)
                if stores and w0:
                    dem = name
                    print(f"{path.split('/')[-1]:22s} {dem[:70]:70s} loop of {b - a:4d} instr: {loads} loads, {stores} stores, {w0} x vmcnt(0)")


if __name__ == "__main__":
    main()
