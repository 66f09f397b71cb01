export ISE_KNN_LIB=$PWD/image-search-engine_amd/csrc/libise_knn_ablate.so
for plan in "8,2" "8,1" "4,2" "4,1" "8,2"; do
  ISE_PLAN=$plan python bench.py --steps 3000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline())
print('plan $plan:', round(r['value']), 'QPS', round(r['ms_per_step']*1e3,1), 'us/step; isolated kernel', round(r['roofline']['kernel_ms']*1e3,1), 'us; power', (r['roofline'].get('board_power') or {}).get('mean_w'))"
done
