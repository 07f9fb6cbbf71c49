#!/bin/bash
# PMC passes over a stand-alone binary (tools/conv_lab.hip, tools/wgrad_lab.hip builds): clock, MFMA busy, L2 hits / misses,
# LDS conflicts, per kernel.   gpurun -- 'bash tools/pmc_bin.sh build/conv_lab 32 5 1 1'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=/tmp/frlw_pmc_bin; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
BIN=$R/$1; shift
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- $BIN $ARGS > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
ARGS="$*"
run a GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
run b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS
run d TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = r["Kernel_Name"][:70] + " grid " + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1]["dur"]))[:8]:
    if "fill" in k or "checksum" in k: continue
    dur = sum(c["dur"]) / len(c["dur"])
    print(f"{k}\n   dur {dur/1e3:.1f} us (under counters)")
    for name in sorted(c):
        if name != "dur": print(f"   {name:32s} {sum(c[name]) / len(c[name]):16.0f}")
PY
