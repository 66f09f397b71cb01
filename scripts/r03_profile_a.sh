#!/bin/bash
# round-3 profile set, part A: bench.py configurations (overlap records, kernel statistics, counters) + bench lines
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
DST=$REPO/gpurun_out/profiles_r03
mkdir -p "$DST"
cd "$REPO"
python bench.py --steps 20 --warmup 5 > "$DST/bench_steps20_warmup5.json" 2> "$DST/bench_steps20_warmup5.err"
scripts/profile_r03.sh n1m_nq16 1000000 16 "void scan_kernel" > "$DST/profile_n1m_nq16.log" 2>&1
scripts/profile_r03.sh n100k_nq16 100000 16 "void short_scan_kernel" > "$DST/profile_n100k_nq16.log" 2>&1
scripts/profile_r03.sh n100k_nq1 100000 1 "void short_scan_kernel" > "$DST/profile_n100k_nq1.log" 2>&1
for a in "--n 100000" "--n 100000 --nq 1" "--n 125000"; do
  tag=$(echo $a | tr -d ' -')
  python bench.py $a --steps 2000 --warmup 50 > "$DST/bench_${tag}.json" 2> "$DST/bench_${tag}.err"
done
python bench.py --steps 20 --warmup 5 > "$DST/bench_steps20_warmup5_with_records.json" 2>> "$DST/bench_steps20_warmup5.err"
ls -la "$DST"
