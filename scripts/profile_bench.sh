#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box: kernel trace + stats, then the
# HBM counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: scripts/profile_bench.sh <tag> [bench args...]
set -eo pipefail
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 50 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name "*.csv" | head -20
