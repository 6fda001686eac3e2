# one-call GPU verification used at the end of the round (through gpurun): new tests first, then timings, then the suite
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dubins" > gpurun_out/dubins_tests.log 2>&1; tail -4 gpurun_out/dubins_tests.log
timeout -k 10 60 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/bench_c5_filter.json 2>gpurun_out/bench_c5_filter.err
RRTX_DUBINS_FILTER=0 timeout -k 10 60 python bench.py --workload c5 --no-cpu-baseline > gpurun_out/bench_c5_nofilter.json 2>gpurun_out/bench_c5_nofilter.err
python - <<'PY'
import json
for f in ("bench_c5_filter", "bench_c5_nofilter"):
    try:
        j = json.load(open("gpurun_out/%s.json" % f)); print(f, j["ms_per_step"], j["value"], j["final_path_cost_mean"], j["paths_found"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout -k 10 280 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_full.log 2>&1; tail -3 gpurun_out/gpu_full.log
