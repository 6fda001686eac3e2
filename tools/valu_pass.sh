#!/bin/bash
# VALU-side roofline of a workload whose kernel is f64-VALU / latency bound (C3..C6): one rocprofv3 --pmc pass
# (counters only) of `bench.py --workload <w>`; writes profiles/r3_<w>_valu.json (read back by bench.py as roofline.valu).
#   RRTX_COMMIT=$(git rev-parse --short HEAD) gpurun -- 'bash tools/valu_pass.sh c5'
W=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/valu_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES \
  --output-format csv -d $OUT -o p -- python3 $REPO/bench.py --workload $W --no-cpu-baseline --warmup 0 --steps 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
echo "valu pass $W rc=$?"
python3 - "$OUT" "$W" "$REPO" <<'PY'
import csv, glob, hashlib, json, os, sys
from collections import defaultdict
out, w, repo = sys.argv[1:4]
s = defaultdict(float); d = defaultdict(int)
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        s[k] += float(row["Counter_Value"]); d[k] += 1
kern = max((k for k, c in s if c == "SQ_INSTS_VALU"), key=lambda k: s[(k, "SQ_INSTS_VALU")])
c = {cn: s[(kern, cn)] for (k, cn) in s if k == kern}
j = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
kms = j["roofline"]["kernel_ms_per_step"]
sys.path.insert(0, os.path.join(repo, "tools"))
import csrc_hash as ch
f64 = c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_TRANS_F64", 0)
other = c["SQ_INSTS_VALU"] - f64
# issue model (MI355X_MICROARCH.md): a wave64 VALU instruction holds its SIMD-32 for 2 cycles, an f64 one for 4
issue_cycles = 2.0 * other + 4.0 * f64
avail = kms * 1e-3 * 2.4e9 * 256 * 4
flops = (2 * c.get("SQ_INSTS_VALU_FMA_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0)) * 64
res = {"workload": w, "kernel": kern, "dispatches": d[(kern, "SQ_INSTS_VALU")], "counters": c, "kernel_ms_profiled_run": kms,
       "valu_insts_per_launch": c["SQ_INSTS_VALU"] / max(d[(kern, "SQ_INSTS_VALU")], 1),
       "valu_busy_frac": issue_cycles / avail,
       "f64_flops_upper_TFLOPs": flops / (kms * 1e-3) / 1e12, "f64_peak_TFLOPs": 78.6,
       "active_lane_frac": c.get("SQ_THREAD_CYCLES_VALU", 0) / max(c.get("SQ_ACTIVE_INST_VALU", 1) * 64.0, 1),
       "f64_valu_note": "valu_busy_frac = (2 cycles x non-f64 VALU instructions + 4 cycles x f64 ones) / (kernel time x 2.4 GHz x "
                        "1024 SIMDs): the share of the chip's VALU issue slots this kernel fills; f64_flops_upper counts 64 lanes per "
                        "instruction (an upper bound: the kernels run many instructions on a few lanes, see active_lane_frac)",
       "csrc_hash": ch.csrc_hash(w), "commit": os.environ.get("RRTX_COMMIT", "unknown")}
json.dump(res, open(os.path.join(repo, "profiles", "r3_%s_valu.json" % w), "w"), indent=1)
json.dump(res, open(os.path.join(repo, "gpurun_out", "r3_%s_valu.json" % w), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("kernel", "valu_busy_frac", "f64_flops_upper_TFLOPs", "active_lane_frac", "kernel_ms_profiled_run")}))
PY
find $OUT -name "*.db" -delete 2>/dev/null
