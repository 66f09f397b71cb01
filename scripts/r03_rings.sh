#!/bin/bash
# chunk size x ring depth of the short-index kernel (dev build), 16-wave blocks; and the crossover against the
# streaming kernel for indexes between 131k and 262k rows (two 8-wave blocks per CU there)
export ISE_KNN_LIB=$PWD/image-search-engine_amd/csrc/libise_knn_ablate.so
run() { python bench.py $2 --steps 2000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('$1 $2: us/step %.1f kernel_us %.1f behind %.1f lat %.1f'%(r['ms_per_step']*1e3, rf['kernel_ms']*1e3, rf['merge_kernel_ms']*1e3, r['batch_latency_us']['median']))"; }
for cfg in "4 2" "4 3" "2 2" "2 3" "2 4" "2 6" "1 4" "1 6"; do
  set -- $cfg
  for a in "--n 100000" "--n 100000 --nq 1"; do
    ISE_SHORT_CH=$1 ISE_SHORT_RING=$2 run "ch $1 ring $2" "$a"
  done
done
for n in 150000 200000 250000; do
  run "short" "--n $n"
  ISE_NO_SHORT=1 run "streaming" "--n $n"
done
