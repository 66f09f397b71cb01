set -o pipefail
REPO=$GRAFT_REPO_ROOT
cd $REPO
timeout -k 10 500 python -m pytest tests/test_exact_l2_gpu.py tests/test_knn_gpu.py -x -q -m gpu -k "gemm or large or bf16" > gpurun_out/t_sel.log 2>&1; tail -2 gpurun_out/t_sel.log
SECONDS_=1 true
SECONDS=60 SEED=77 timeout -k 10 200 python scripts/fuzz_gemm_probe.py > gpurun_out/fz_sel.log 2>&1; tail -2 gpurun_out/fz_sel.log
cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_sel; rm -rf "$OUT"; mkdir -p "$OUT"
STORAGE=bf16 NQS=1024 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/gemm_probe.py" child > "$OUT/trace.log" 2>&1
grep "nq, ms" "$OUT/trace.log"
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
