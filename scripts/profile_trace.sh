#!/bin/bash
# one rocprofv3 kernel-trace + stats pass of bench.py; usage: scripts/profile_trace.sh <tag> [bench args...]
set -eo pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" | head -3
