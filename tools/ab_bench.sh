#!/bin/bash
# Same-box A/B of librrtx builds (boxes differ by up to 9 % for one binary: only pairs measured in one gpurun call count).
#   gpurun -- 'bash tools/ab_bench.sh "c2 c5" librrtx.so librrtx_base.so ...'   (libraries under robotics-path-planning_amd/)
# AB_ARGS="--instances 2048" is handed to bench.py.  Two interleaved repetitions per (workload, library); one line each: ms per step and the bench value.
WL="$1"; shift
mkdir -p gpurun_out
for w in $WL; do
  steps=1; [ "$w" != c2 ] && steps=3
  for rep in 1 2; do
    for lib in "$@"; do
      out=gpurun_out/ab_${w}_${lib}_${rep}.json
      RRTX_LIB=$PWD/robotics-path-planning_amd/$lib timeout -k 10 200 python3 bench.py --workload $w --steps $steps --warmup 1 \
        --no-cpu-baseline $AB_ARGS > $out 2> gpurun_out/ab_err.txt || { echo "FAILED $w $lib"; tail -3 gpurun_out/ab_err.txt; exit 1; }
      python3 - "$out" "$w" "$lib" "$rep" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s %-22s rep %s  %9.1f ms/step  value %.4g  cost %s" % (sys.argv[2], sys.argv[3], sys.argv[4], j["ms_per_step"], j["value"],
      j.get("final_path_cost_mean")), flush=True)
PY
    done
  done
done
