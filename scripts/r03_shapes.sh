#!/bin/bash
# throughput / latency of bench.py at short sizes by block shape of the short-index kernel (dev build)
export ISE_KNN_LIB=$PWD/image-search-engine_amd/csrc/libise_knn_ablate.so
for shape in 0 1 2; do
  for ring in 2 3; do
    for a in "--n 100000" "--n 100000 --nq 1" "--n 125000"; do
      ISE_SHORT_SHAPE=$shape ISE_SHORT_RING=$ring python bench.py $a --steps 2000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('shape $shape ring $ring $a: us/step %.1f kernel_us %.1f behind %.1f lat %.1f'%(r['ms_per_step']*1e3, rf['kernel_ms']*1e3, rf['merge_kernel_ms']*1e3, r['batch_latency_us']['median']))"
    done
  done
done
