# One-call GPU verification used at the end of a round (through gpurun): the driver's exact bench command under its
# own 600 s limit first, then the GPU suite.
#   gpurun --timeout 1100 -- 'bash tools/round_end_check.sh'
mkdir -p gpurun_out
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
echo "bench rc=$? $(tail -c 400 gpurun_out/bench_driver.err)"
python3 - <<'PY'
import json
try:
    j = json.loads(open("gpurun_out/bench_driver.json").read().strip().splitlines()[-1])
    print("steps", j["steps"], "ms_per_step", j["ms_per_step"], "value", j["value"], "frac", j["roofline"]["frac"],
          "cpu", (j.get("cpu_baseline") or {}).get("value"), "elapsed", j["elapsed_s"])
except Exception as e:
    print("bench line ERR", e)
PY
timeout -k 10 420 python3 -m pytest tests -x -q -m gpu > gpurun_out/gpu_full.log 2>&1; tail -3 gpurun_out/gpu_full.log
