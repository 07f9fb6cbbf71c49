#!/bin/bash
# Effective shader clock and MFMA-busy share of every kernel of a stand-alone binary (tools/conv_lab.hip builds).
#   gpurun -- 'bash tools/pmc_clock_bin.sh build/conv_lab 32 20 2'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_pmc_clock_bin; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BIN=$R/$1; shift
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/a -o p -- $BIN "$@" > $OUT/a.log 2>&1; echo rc=$?
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = (n[:60] + " grid " + r.get("Grid_Size", "?"))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur"]))[:12]:
    n = len(c["GRBM_GUI_ACTIVE"]); dur = sum(c["dur"]) / len(c["dur"])
    g = sum(c["GRBM_GUI_ACTIVE"]) / n / 8
    m = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(1, len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [1])))
    print(f"{k:90s} dur {dur/1e3:8.1f} us  clock {g/dur:5.2f} GHz  mfma busy {m/1024/max(g,1)*100:5.1f} % of active cycles")
PY
