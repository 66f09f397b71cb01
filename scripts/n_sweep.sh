for n in 25000 50000 100000 125000 150000 200000 400000; do
  python bench.py --n $n --steps 4000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.readline()); n = r['config']['n']
print(n, 'rows', round(n*512*4/1e6), 'MB:', round(r['ms_per_step']*1e3,1), 'us/step ->', round(r['roofline']['step_effective']['achieved']), 'GB/s effective; isolated kernel', round(r['roofline']['kernel_ms']*1e3,1), 'us')"
done
