#!/bin/bash
# rocprofv3 --kernel-trace --stats of the secondary bench workloads (c3, c4, c5, c6; WORKLOADS="c6" selects) on the GPU box; summaries are copied
# to gpurun_out/ as <tag>_<workload>_kernel_stats.csv.  Usage (through gpurun): bash tools/profile_workloads.sh <tag>
set -e
TAG=${1:-r1}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for W in ${WORKLOADS:-c3 c4 c5 c6}; do
  OUT=$REPO/gpurun_out/prof_${TAG}_$W
  mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python3 $REPO/bench.py --workload $W --no-cpu-baseline --warmup 0 --steps 1 > $OUT/bench.json 2> $OUT/err.txt
  cp $OUT/ks_kernel_stats.csv $REPO/gpurun_out/${TAG}_${W}_kernel_stats.csv
  echo "$W done"
done
