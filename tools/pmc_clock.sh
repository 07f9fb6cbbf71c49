#!/bin/bash
# Effective shader clock of every kernel = GRBM_GUI_ACTIVE / 8 (XCDs) / kernel duration (MI355X_MICROARCH.md, DVFS).
#   gpurun -- 'bash tools/pmc_clock.sh tools/det_layers.py'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_pmc_clock; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
SCRIPT=$R/$1; shift
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -o p -- python3 $SCRIPT "$@" > $OUT/a.log 2>&1; echo rc=$?
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
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
