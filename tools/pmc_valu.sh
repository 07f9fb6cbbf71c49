#!/bin/bash
# VALU / SALU / LDS instruction counts and wave cycles per kernel of one enc_lab configuration (one rocprofv3 --pmc pass)
#   gpurun -- 'bash tools/pmc_valu.sh <cfg> [lib.so]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; CFG=${1:-mpx}; LIB=${2:-$R/frlw-evd_amd/csrc/libfrlw_evd.so}
OUT=/tmp/frlw_pmc_valu; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $OUT -o p -- $R/build/enc_lab $LIB --cfg $CFG --reps 3 > $OUT/log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    if "rocclr" in k or "selftest" in k or "leaky" in k: continue
    w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
    print(f"{k:42s} waves {w:8.0f}  " + "  ".join(f"{n[3:]} {sum(v)/len(v)/1e6:7.2f}M ({sum(v)/len(v)/w:6.0f}/wave)" for n, v in sorted(c.items()) if n != "SQ_WAVES"))
PY
