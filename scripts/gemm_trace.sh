#!/bin/bash
# rocprofv3 kernel trace of the large-batch path at 1M x 512, nq = 1024
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_${1:-gemm}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
NQS=${NQS:-1024} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/gemm_probe.py" child > "$OUT/trace.log" 2>&1
for f in "$OUT"/trace/*/*kernel_stats.csv; do cut -c1-150 "$f" | head -12; done
