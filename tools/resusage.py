#!/usr/bin/env python3
"""Compiler view of every kernel in a .hip file: VGPRs, scratch, LDS, occupancy."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
       "-I", "include", "-I", "frlw-evd_amd/csrc", "-c", src, "-o", "/tmp/_ru.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    s = m.group(1)
    if s.startswith("Function Name:"):
        name = s.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|frlw::|void ", "", name).split("(")[0]
        rows[cur] = {}
    elif cur and ":" in s:
        k, v = s.split(":", 1)
        rows[cur][k.strip().split(" [")[0]] = v.strip()
for k, r in rows.items():
    print(f"{k[:44]:44s} vgpr={r.get('VGPRs','?'):>4s} scratch={r.get('ScratchSize','?'):>5s} lds={r.get('LDS Size','?'):>6s} occ={r.get('Occupancy','?')}")
