#!/bin/bash
# kernel time of gemm_scan_bf16_kernel alone under the dev build's phase switches (rocprofv3 kernel stats)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export ISE_KNN_LIB=$REPO/image-search-engine_amd/csrc/libise_knn_ablate.so
cd /tmp && export TMPDIR=/tmp
for abl in ${ABLS:-0 1 2 3 4 5 6 7}; do
  OUT=$REPO/gpurun_out/prof_gbabl_$abl
  rm -rf "$OUT"; mkdir -p "$OUT"
  ISE_GEMM_ABLATE=$abl STORAGE=bf16 NQS=1024 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/gemm_probe.py" child > "$OUT/trace.log" 2>&1
  f=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$abl" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("void gemm_scan_bf16_kernel<16, false"):
        print(f"ablate {sys.argv[2]}: main pass {float(r['AverageNs'])/1e3:8.1f} us avg over {r['Calls']} launches")
PY
  rm -rf "$OUT"
done
