#!/bin/bash
# throughput of bench.py at 100k rows by number of HIP hardware queues and by path
for q in 4 8 16; do
  for ns in "" 1; do
    for nq in 16 1; do
      GPU_MAX_HW_QUEUES=$q ISE_NO_SHORT=$ns python bench.py --n 100000 --nq $nq --steps 2000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); rf=r['roofline']
print('queues $q no_short=${ns:-0} nq=$nq: us/step %.1f kernel_us %.1f behind %.1f lat %.1f'%(r['ms_per_step']*1e3, rf['kernel_ms']*1e3, rf['merge_kernel_ms']*1e3, r['batch_latency_us']['median']))"
    done
  done
done
