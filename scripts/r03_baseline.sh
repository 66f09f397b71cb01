#!/bin/bash
# round-3 baseline: the short-index configurations measured the way the headline is (bench.py, 16 streams)
set -eo pipefail
OUT=gpurun_out/${1:-r03_base}
mkdir -p $OUT
python bench.py --n 100000 --steps 2000 --warmup 50 --no-cpu-baseline > $OUT/n100k_nq16.json 2> $OUT/n100k_nq16.err
python bench.py --n 100000 --nq 1 --steps 2000 --warmup 50 --no-cpu-baseline > $OUT/n100k_nq1.json 2> $OUT/n100k_nq1.err
python bench.py --n 125000 --steps 2000 --warmup 50 --no-cpu-baseline > $OUT/n125k_nq16.json 2> $OUT/n125k.err
cat $OUT/*.json
