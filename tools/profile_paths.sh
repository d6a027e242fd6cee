#!/bin/bash
# rocprofv3 passes over tools/bench_paths.py (the non-headline rows: cascade, DLT, next rows):
# one --kernel-trace --stats run and two separate --pmc passes, CSV under gpurun_out/prof_<tag>/.
#   tools/profile_paths.sh <tag> [bench_paths.py arguments, e.g. --only cascade,dlt]
# Summarise with tools/pmc_kernels.py.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
shift
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/bench_paths.py" $ARGS > "$OUT/stats.jsonl" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/bench_paths.py" $ARGS > "$OUT/fetch.jsonl" 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/bench_paths.py" $ARGS > "$OUT/write.jsonl" 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
