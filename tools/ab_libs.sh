#!/bin/bash
# Same-box A/B of library builds (GPU box):  gpurun -- 'bash tools/ab_libs.sh c3 "<bench args>" libA.so libB.so ...'
# Runs `bench.py --workload <w>` once per library (RRTX_LIB), in the order given and then once more in reverse order
# (boxes differ by up to 9 %, and the first run of a call pages the image in: only same-box pairs count), and prints
# ms_per_step / kernel ms / final cost per run.  Output under gpurun_out/ab/.
W=$1; ARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/ab
mkdir -p $O
cd $REPO
LIBS=("$@")
ORDER=("${LIBS[@]}")
for ((i=${#LIBS[@]}-1; i>=0; i--)); do ORDER+=("${LIBS[$i]}"); done
k=0
for L in "${ORDER[@]}"; do
  k=$((k+1))
  RRTX_LIB=$REPO/robotics-path-planning_amd/$L timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline $ARGS > $O/run_$k.json 2> $O/run_$k.err || { echo "run $k ($L) failed"; tail -3 $O/run_$k.err; exit 1; }
  python3 - "$O/run_$k.json" "$L" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("%-28s ms/step %9.2f  kernel ms %9.2f  cost %.12f  value %.4g" % (sys.argv[2], j["ms_per_step"], r["kernel_ms_per_step"], j["final_path_cost_mean"] or 0.0, j["value"]))
PY
done
