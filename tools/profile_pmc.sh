#!/bin/bash
# One rocprofv3 --pmc pass (counters only, no trace domain) over tools/bench_paths.py.
#   tools/profile_pmc.sh <tag> "<COUNTER COUNTER ...>" [bench_paths.py arguments]
# (counters of ONE hardware block per pass: a set the hardware cannot collect together makes rocprofv3
# abort and then sit until killed -- hence the timeout)
# CSV lands in gpurun_out/pmc_<tag>/; tools/pmc_table.py turns it into a per-kernel table.
set -eo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
COUNTERS=$2
shift 2
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --pmc $COUNTERS --output-format csv -d "$OUT/run" -- python3 "$ROOT/tools/bench_paths.py" --steps 3 --warmup 1 "$@" > "$OUT/out.jsonl" 2> "$OUT/err.log"
find "$OUT" -name '*.db' -delete
echo "pmc pass $TAG done"
