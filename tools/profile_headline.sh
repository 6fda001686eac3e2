#!/bin/bash
# Profile the headline bench on the GPU box: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate
# PMC passes (never combined with a trace domain).  Usage (through gpurun): bash tools/profile_headline.sh <tag> [bench args]
set -e
TAG=${1:-r1_f32}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o ks -- python3 $REPO/bench.py --no-cpu-baseline --warmup 0 --steps 1 "$@" > $OUT/bench_ks.json 2> $OUT/ks.err
echo "kernel-trace pass done" 
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $REPO/bench.py --no-cpu-baseline --warmup 0 --steps 1 "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $REPO/bench.py --no-cpu-baseline --warmup 0 --steps 1 "$@" > $OUT/bench_write.json 2> $OUT/write.err
echo "WRITE_SIZE pass done"
# keep only the summaries (the per-dispatch CSVs are small; drop big traces)
find $OUT -name "*.db" -delete 2>/dev/null || true
python3 $REPO/tools/pmc_summary.py $OUT $TAG $SUMMARY_ARGS
cp $REPO/profiles/${TAG}_* $REPO/gpurun_out/ 2>/dev/null || true
