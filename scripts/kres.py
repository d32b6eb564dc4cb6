#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table for one csrc file:  python scripts/kres.py conv_gemm.hip [filter]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "whisper-vits-svc_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", sys.argv[1], "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True).stderr
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        cur = {"name": name}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
print(f"{'kernel':48s} {'sgpr':>5s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>7s} {'occ':>4s} {'lds':>7s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:48]:48s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} "
              f"{r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>4s} {r.get('LDS','?'):>7s}")
