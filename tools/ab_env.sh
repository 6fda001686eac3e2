#!/bin/bash
# Same-box A/B of one environment switch of librrtx.so (boxes differ by several per cent for one binary: only pairs measured
# in one gpurun call count).   gpurun -- 'bash tools/ab_env.sh c2 RRTX_SPEC2 1 0'   (AB_ARGS / AB_STEPS as in ab_bench.sh)
# Two interleaved repetitions per value; one line each: ms per step, the bench value and the mean path cost.
W="$1"; VAR="$2"; shift 2
mkdir -p gpurun_out
steps=${AB_STEPS:-1}
for rep in 1 2; do
  for val in "$@"; do
    out=gpurun_out/abenv_${W}_${VAR}_${val}_${rep}.json
    env "$VAR=$val" timeout -k 10 300 python3 bench.py --workload $W --steps $steps --warmup 1 --no-cpu-baseline $AB_ARGS > $out 2> gpurun_out/ab_err.txt \
      || { echo "FAILED $W $VAR=$val"; tail -3 gpurun_out/ab_err.txt; exit 1; }
    python3 - "$out" "$W" "$VAR=$val" "$rep" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j.get("roofline", {})
print("%s %-16s rep %s  %9.1f ms/step  value %.4g  cost %s  frac %s  shared %s" % (sys.argv[2], sys.argv[3], sys.argv[4], j["ms_per_step"],
      j["value"], j.get("final_path_cost_mean"), r.get("frac"), j.get("passes_shared_last_step")), flush=True)
PY
  done
done
