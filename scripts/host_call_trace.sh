#!/bin/bash
# rocprofv3 kernel statistics of the reference's own call (scripts/host_call_probe.py at 1000 x 2048, k = 20), both
# index types -> gpurun_out/host_call_trace/{l2,ip}_kernel_stats.csv
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/host_call_trace; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export CASES="1000,2048,20"
for m in l2 ip; do
  METRIC=$m rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$m" -- python3 "$REPO/scripts/host_call_probe.py" > "$OUT/$m.log" 2>&1
  f=$(find "$OUT/$m" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${m}_kernel_stats.csv"
  grep "k=" "$OUT/$m.log"
done
