#!/bin/bash
set -o pipefail
bash scripts/collect_traffic.sh r03b latest > gpurun_out/traffic_r03b.log 2>&1; echo "traffic k31 rc=$?"; tail -3 gpurun_out/traffic_r03b.log
bash scripts/collect_traffic.sh r03b_k63 --k 63 > gpurun_out/traffic_r03b_k63.log 2>&1; echo "traffic k63 rc=$?"; tail -3 gpurun_out/traffic_r03b_k63.log
mkdir -p gpurun_out/profiles_out; cp profiles/traffic_r03b*.json profiles/traffic_latest.json gpurun_out/profiles_out/ 2>/dev/null
sed -e 's/r03_sq_counters/r03b_sq_counters/g' -e 's#gpurun_out/sq_r03#gpurun_out/sq_r03b#' scratch/sq_counters_r03.sh > /tmp/sq.sh; bash /tmp/sq.sh > gpurun_out/sq_r03b.log 2>&1; echo "sq rc=$?"; cat gpurun_out/r03b_sq_counters.txt
