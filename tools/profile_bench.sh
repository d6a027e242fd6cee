#!/bin/bash
# Profiles `bench.py` on the GPU box the way the measurement contract asks: one
# `rocprofv3 --kernel-trace --stats` run and two separate `--pmc` passes (FETCH_SIZE, WRITE_SIZE;
# never combined with a trace domain), CSV output under gpurun_out/prof_<tag>/.
#
#   tools/profile_bench.sh <tag> [extra bench.py arguments]
#
# Afterwards (here or on the build host, from the merged gpurun_out/):
#   python tools/pmc_summary.py --stats-dir gpurun_out/prof_<tag>/stats --fetch-dir gpurun_out/prof_<tag>/fetch \
#       --write-dir gpurun_out/prof_<tag>/write --tag <tag> --xrows ... --yrows ...
# writes profiles/<tag>_kernel_stats.md and the per-shape record in profiles/l1k2_pmc.json.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
shift
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 3 --warmup 1 --cpu-seconds 0 --verify 0 $*"
# the program itself after `--` (no env/bash hop: the profiler has initialised the GPU already)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch.json" 2> "$OUT/fetch.err"
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write.json" 2> "$OUT/write.err"
echo "WRITE_SIZE pass done"
# keep what travels back small: the per-dispatch CSVs are all the summary needs
find "$OUT" -name '*.db' -delete
du -sh "$OUT"
