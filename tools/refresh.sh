#!/bin/bash
# Round-end refresh of everything under profiles/ (run on the GPU box: gpurun -- 'bash tools/refresh.sh r01').
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/refresh_$TAG
mkdir -p "$O"
cd "$R"
python3 bench.py > "$O/bench.json" 2> "$O/bench.log"; echo "bench rc=$?"; tail -c 1500 "$O/bench.json"
python3 bench.py --hotspot --no-detector --no-train --no-cpu-baseline --steps 10 > "$O/bench_hotspot.json" 2>> "$O/bench.log"
bash tools/profile.sh ${TAG}_taf --no-detector --no-train --no-also > "$O/prof_taf.log" 2>&1
bash tools/profile.sh ${TAG}_hot --hotspot --no-detector --no-train --no-also --steps 5 > "$O/prof_hot.log" 2>&1
bash tools/profile.sh ${TAG}_gen1 --workload taf_gen1 --no-detector --no-train > "$O/prof_gen1.log" 2>&1
bash tools/profile.sh ${TAG}_det --no-train --no-also --steps 3 > "$O/prof_det.log" 2>&1
EXTRA="--no-detector --no-train --no-also" bash tools/pmc.sh $TAG > "$O/pmc.log" 2>&1
cp gpurun_out/pmc_$TAG/summary.txt "$O/pmc_summary.txt" 2>/dev/null
for t in taf hot gen1 det; do
  cp gpurun_out/prof_${TAG}_$t/kernel_stats.csv "$O/${t}_kernel_stats.csv" 2>/dev/null
done
ls -la "$O"
