set -o pipefail
cd $GRAFT_REPO_ROOT
D=gpurun_out/final_caps; mkdir -p $D
export CASES="1000,2048,20;10000,2048,20;1000,512,20;10000,512,20;1000,128,9;100000,512,10;1000000,512,10"
{ echo "# scripts/host_call_probe.py, IndexFlatL2"; python scripts/host_call_probe.py 2>&1 | grep "k="; echo "# METRIC=ip (IndexFlatIP)"; METRIC=ip python scripts/host_call_probe.py 2>&1 | grep "k="; } > $D/host_call_latency.txt
{ echo "# ISE_DIRECT_SHORT_MAX_TILES=100000 scripts/direct_crossover_probe.py (back to back)"; ISE_DIRECT_SHORT_MAX_TILES=100000 python scripts/direct_crossover_probe.py 2>&1 | grep "direct"; } > $D/direct_crossover.txt
python bench.py --steps 20 --warmup 5 > $D/bench_steps20_warmup5_with_records.json 2> $D/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > $D/smoke.log 2>&1; tail -2 $D/smoke.log
scripts/profile_r03.sh n100k_nq1 100000 1 "void short_scan_kernel" > $D/profile_n100k_nq1.log 2>&1
cp gpurun_out/profiles_r03/bench_n100000_nq1_* $D/
cat $D/host_call_latency.txt
