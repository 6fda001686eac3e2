#!/bin/bash
# One rocprofv3 --pmc pass of bench.py (counters only: never combined with a trace domain), summed per kernel.
#   bash tools/pmc_pass.sh <tag> "<COUNTER ...>" [bench args]      -> gpurun_out/pmc_<tag>.txt
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o p -- python3 $REPO/bench.py --no-cpu-baseline --warmup 0 --steps 1 "$@" > $OUT/bench.json 2> $OUT/err.txt
echo "pmc pass $TAG rc=$?"
python3 - "$OUT" > $REPO/gpurun_out/pmc_$TAG.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
s = defaultdict(float); d = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
        s[k] += float(row["Counter_Value"]); d[k] += 1
for (k, c), v in sorted(s.items()):
    print("%-48s %-28s dispatches %5d  sum %.6g" % (k, c, d[(k, c)], v))
PY
find $OUT -name "*.db" -delete 2>/dev/null
cat $REPO/gpurun_out/pmc_$TAG.txt | grep -v "init_kernel\|root_kernel\|copyBuffer"
