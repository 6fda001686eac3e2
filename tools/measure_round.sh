#!/bin/bash
# Round-end measurement set (GPU box):  gpurun --timeout 1200 -- 'RRTX_COMMIT=<short hash> bash tools/measure_round.sh a'   then   '... b'
# (two calls: one call may run 1 200 s at most.  a = C2: profile passes, the driver's command, phase splits; b = the other workloads)
# (the box has no .git: the hash of the commit being measured is handed in; do not edit the tree while the call is queued --
# the snapshot is taken when the call starts)
#   1. rocprofv3 passes of the headline bench: --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes)
#      -> profiles/r3_c2_kernel_stats.csv, r3_c2_pmc_*.csv, r3_c2_traffic.json (device-code hash + commit inside)
#   2. the driver's exact bench command under its 600 s limit -> r3_bench_c2_driver_cmd.json (reads the traffic file of 1.)
#   3. per-phase split of the C2 iteration (diagnostic build) -> r3_c2_phase_4096x105k.txt; the same for the rrt_07 / rrt_05 kernels
#      (needs robotics-path-planning_amd/librrtx_prof.so: `make -C robotics-path-planning_amd/csrc prof`, and tools/ubench/lat_ubench)
#   4. C3..C6: bench line + VALU PMC pass -> r3_bench_<w>.json, r3_<w>_valu.json
# Everything is written under gpurun_out/r3/ (merged back by gpurun); copy it into profiles/ and commit.
export RRTX_COMMIT=${RRTX_COMMIT:-unknown}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/r3
mkdir -p $O
cd $REPO
PART=${1:-ab}
if [[ $PART == *a* ]]; then
SUMMARY_ARGS="--workload c2" bash tools/profile_headline.sh r3_c2 > $O/profile_headline.log 2>&1
cp profiles/r3_c2_* $O/ 2>/dev/null
tail -2 $O/profile_headline.log
# cpu_fullsize first so that the driver-command line carries the full-size CPU figure
timeout -k 10 400 python3 tools/cpu_fullsize.py c2 > $O/cpu_fullsize_c2.log 2>&1
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r3_bench_c2_driver_cmd.json 2> $O/bench_driver.err
echo "driver bench rc=$?"
RRTX_LIB=robotics-path-planning_amd/librrtx_prof.so timeout -k 10 120 python3 tools/phase_profile.py 4096 105000 > $O/r3_c2_phase_4096x105k.txt 2>&1
echo "phase rc=$?"
RRTX_LIB=robotics-path-planning_amd/librrtx_prof.so timeout -k 10 120 python3 tools/phase_profile_c3.py 1024 20000 > $O/r3_c3_phase.txt 2>&1
RRTX_LIB=robotics-path-planning_amd/librrtx_prof.so timeout -k 10 120 python3 tools/phase_profile_c5.py 1536 5000 > $O/r3_c5_phase.txt 2>&1
echo "phase c3/c5 rc=$?"
timeout -k 10 200 python3 bench.py --instances 8192 --warmup 1 --steps 1 --no-cpu-baseline > $O/r3_bench_c2_8192.json 2> $O/bench_c2_8192.err
RRTX_SPEC2=0 timeout -k 10 200 python3 bench.py --warmup 1 --steps 2 --no-cpu-baseline > $O/r3_bench_c2_one_pass_per_iteration.json 2> $O/bench_c2_spec0.err
echo "extra c2 benches rc=$?"
fi
if [[ $PART == *b* ]]; then
# C3 (rrt_07): kernel stats + HBM traffic passes of its own (profiles/r3_c3_kernel_stats.csv, r3_c3_traffic.json)
SUMMARY_ARGS="--workload c3 --kernel-match rrt_informed_kernel --instances 1024 --max-iter 20000 --obstacles 200 --variant q16_mirror" bash tools/profile_headline.sh r3_c3 --workload c3 > $O/profile_headline_c3.log 2>&1
cp profiles/r3_c3_* $O/ 2>/dev/null
tail -1 $O/profile_headline_c3.log
# kernel stats of the other workloads (C4: the bounded BIT* launches; C5, C6)
WORKLOADS="c4 c5 c6" bash tools/profile_workloads.sh r3 > $O/profile_workloads.log 2>&1
cp gpurun_out/r3_c?_kernel_stats.csv $O/ 2>/dev/null
# full-size single-thread CPU baseline (cached by CPU model; bench.py's cpu_baseline reports it)
timeout -k 10 400 python3 tools/cpu_fullsize.py c2 c3 c5 > $O/cpu_fullsize.log 2>&1
cp profiles/cpu_fullsize_*.json $O/ 2>/dev/null
echo "cpu fullsize rc=$?"
# C4 at 16 384 instances (work queue, longest expected run first; 5 walled-in starts end RRTX_ST_REF_HANGS) and C2 at 8 192
timeout -k 10 150 python3 bench.py --workload c4 --instances 16384 --warmup 1 --steps 5 --no-cpu-baseline > $O/r3_bench_c4_16384.json 2> $O/bench_c4_16384.err
RRTX_BITSTAR_FIFO=1 timeout -k 10 150 python3 bench.py --workload c4 --instances 16384 --warmup 1 --steps 5 --no-cpu-baseline > $O/r3_bench_c4_16384_fifo.json 2> $O/bench_c4_16384_fifo.err
echo "extra benches rc=$?"
for w in c3 c4 c5 c6; do
  bash tools/valu_pass.sh $w > $O/valu_$w.log 2>&1
  cp gpurun_out/r3_${w}_valu.json $O/ 2>/dev/null
  cp gpurun_out/r3_${w}_valu.json profiles/ 2>/dev/null
  timeout -k 10 150 python3 bench.py --workload $w --warmup 1 --steps 5 > $O/r3_bench_$w.json 2> $O/bench_$w.err
  echo "bench $w rc=$?"
done
timeout -k 10 120 tools/ubench/lat_ubench > $O/r3_lat_ubench.txt 2>&1
fi
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3/r3_bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); r = j["roofline"]
        print(f.split("/")[-1], "steps", j["steps"], "ms/step %.1f" % j["ms_per_step"], "value %.4g" % j["value"],
              "bound", r["bound"], "frac", r["frac"], "traffic_frac", r.get("traffic_frac"), "elapsed %.0f" % j["elapsed_s"])
    except Exception as e:
        print(f, "ERR", e)
PY
