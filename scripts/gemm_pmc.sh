#!/bin/bash
# SQ counters of the bf16 GEMM-shaped pass at 1M x 512, nq = 1024 (one --pmc pass, no tracing beside it)
set -eo pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_${1:-gemm_pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
STORAGE=${STORAGE:-bf16} NQS=${NQS:-1024} rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc1" -- python3 "$REPO/scripts/gemm_probe.py" child > "$OUT/pmc1.log" 2>&1
STORAGE=${STORAGE:-bf16} NQS=${NQS:-1024} rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc2" -- python3 "$REPO/scripts/gemm_probe.py" child > "$OUT/pmc2.log" 2>&1 || true
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
out = sys.argv[1]
for sub in ("pmc1", "pmc2"):
    for f in glob.glob(f"{out}/{sub}/*/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "gemm_scan" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(sub, k)
            for c, vals in sorted(v.items()):
                print(f"   {c:34s} mean {sum(vals)/len(vals):16.1f}  (n={len(vals)})")
PY
