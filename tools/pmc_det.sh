#!/bin/bash
# MFMA / issue counters of the detector convolutions.  gpurun -- 'bash tools/pmc_det.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_pmc_det; rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/pmc_det; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/a -o p -- python3 $R/tools/det_layers.py > $OUT/a.log 2>&1; echo rc=$?
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -o p -- python3 $R/tools/det_layers.py > $OUT/b.log 2>&1; echo rc=$?
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_conv_mfma" not in n: continue
        key = n[n.index("<"):n.index(">")+1] + " grid " + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    tot = {n: sum(v) for n, v in c.items()}
    print(k, {n: int(v / len(c[n])) for n, v in tot.items()})
PY
