#!/bin/bash
# Per-kernel VGPR / scratch / LDS / occupancy of one .hip file (compiler view).
F=$1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-fast-math -I /root/repo/include \
  -I /root/repo/frlw-evd_amd/csrc -c "$F" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c '
import re,sys,subprocess
cur=None
for line in sys.stdin:
    m=re.search(r"remark: .*?: (.*?) \[-Rpass", line)
    if not m: continue
    s=m.group(1).strip()
    if s.startswith("Function Name:"):
        name=s.split(":",1)[1].strip()
        try: name=subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt",name]).decode().strip()
        except Exception: pass
        name=re.sub(r"\(anonymous namespace\)::|frlw::|void ","",name).split("(")[0]
        print("\n"+name[:50].ljust(50),end=" ")
    elif any(s.startswith(k) for k in ("VGPRs:","AGPRs","ScratchSize","Occupancy","LDS Size","SGPRs:","VGPR Spill")):
        print(s.replace(" [bytes/lane]","").replace(" [bytes/block]","").replace(" [waves/SIMD]",""),end=" | ")
print()
'
