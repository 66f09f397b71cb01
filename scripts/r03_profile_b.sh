#!/bin/bash
# round-3 profile set, part B: one-query batches at 1M, the 125k shard, large batches, config 4's assignment kernel
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
DST=$REPO/gpurun_out/profiles_r03
mkdir -p "$DST"
cd "$REPO"
scripts/profile_r03.sh n1m_nq1 1000000 1 "void exact_scan_kernel<1" > "$DST/profile_n1m_nq1.log" 2>&1
scripts/profile_r03.sh n125k_nq16 125000 16 "void short_scan_kernel" > "$DST/profile_n125k_nq16.log" 2>&1
python bench.py --nq 1 --steps 200 --warmup 20 --no-cpu-baseline > "$DST/bench_nq1.json" 2> "$DST/bench_nq1.err"
python bench.py --nq 1024 --steps 20 --warmup 5 --no-cpu-baseline > "$DST/bench_nq1024.json" 2> "$DST/bench_nq1024.err"
ISE_BENCH_FORCE_SHARDED=1 scripts/profile_trace.sh r03_sh125k --n 125000 --steps 400 --warmup 40 > /dev/null 2>&1
cp "$(find gpurun_out/prof_r03_sh125k -name '*kernel_stats.csv' | head -1)" "$DST/sharded_rehearsal_world1_125k_kernel_stats.csv"
cd /tmp && export TMPDIR=/tmp
for cfg in "f32_l2 f32" "bf16_ip bf16"; do
  set -- $cfg
  OUT=$REPO/gpurun_out/prof_gemm_$1; rm -rf "$OUT"; mkdir -p "$OUT"
  STORAGE=$2 NQS=1024 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/gemm_probe.py" child > "$OUT/trace.log" 2>&1
  cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$DST/gemm_$1_nq1024_kernel_stats.csv"
  grep "nq, ms" "$OUT/trace.log" > "$DST/gemm_$1_nq1024.txt"
  rm -rf "$OUT"
done
OUT=$REPO/gpurun_out/prof_gemm_f32_ip; rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/ip_gemm_probe.py" > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$DST/gemm_f32_ip_kernel_stats.csv"
grep "nq=" "$OUT/trace.log" > "$DST/gemm_f32_ip.txt"; rm -rf "$OUT"
OUT=$REPO/gpurun_out/prof_assign; rm -rf "$OUT"; mkdir -p "$OUT"
N=50000000 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/assign_probe.py" > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$DST/assign_config4_50M_kernel_stats.csv"
grep "assign" "$OUT/trace.log" > "$DST/assign_config4_50M.txt"; rm -rf "$OUT"
ls -la "$DST"
