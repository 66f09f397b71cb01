#!/bin/bash
# Round-3 profile set for one bench.py configuration, on the GPU box:
#   1. kernel trace of the TIMED configuration (16 issue streams) -> overlap record (scripts/overlap_from_trace.py)
#   2. kernel trace + stats on ONE stream (isolated kernel durations)
#   3. FETCH_SIZE and WRITE_SIZE in their own passes (one stream) -> traffic record (scripts/pmc_to_json.py)
# usage: scripts/profile_r03.sh <tag> <n> <nq> <kernel-name-prefix> [extra bench args]
# results: gpurun_out/profiles_r03/<tag>_*  (copy what is to be judged into profiles/r03/)
set -eo pipefail
TAG=$1; N=$2; NQ=$3; KPREFIX=$4; shift 4 || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
DST=$REPO/gpurun_out/profiles_r03
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
ARGS="--n $N --nq $NQ --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace16" -- python3 "$REPO/bench.py" $ARGS --steps 2000 --warmup 50 > "$OUT/trace16.log" 2>&1
python3 "$REPO/scripts/overlap_from_trace.py" "$(find "$OUT/trace16" -name '*kernel_trace.csv' | head -1)" --kernel "$KPREFIX" --last 1500 \
    --json "$DST/bench_n${N}_nq${NQ}_streams16_overlap.json" > /dev/null
cp "$(find "$OUT/trace16" -name '*kernel_stats.csv' | head -1)" "$DST/bench_n${N}_nq${NQ}_streams16_kernel_stats.csv"
ISE_BENCH_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace1" -- python3 "$REPO/bench.py" $ARGS --steps 500 --warmup 20 > "$OUT/trace1.log" 2>&1
cp "$(find "$OUT/trace1" -name '*kernel_stats.csv' | head -1)" "$DST/bench_n${N}_nq${NQ}_streams1_kernel_stats.csv"
ISE_BENCH_STREAMS=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS --steps 50 --warmup 5 > "$OUT/pmc_fetch.log" 2>&1
ISE_BENCH_STREAMS=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS --steps 50 --warmup 5 > "$OUT/pmc_write.log" 2>&1
python3 "$REPO/scripts/pmc_to_json.py" "$OUT" "$DST/bench_n${N}_nq${NQ}_hbm_pmc.json" "$N" "$NQ" "$KPREFIX" > "$OUT/pmc_to_json.log" 2>&1 || echo "pmc_to_json failed for $TAG (see $OUT/pmc_to_json.log)"
rm -rf "$OUT/trace16" "$OUT/trace1" "$OUT/pmc_fetch" "$OUT/pmc_write"   # the raw traces are large; the reductions are kept
ls -la "$DST" | tail -8
