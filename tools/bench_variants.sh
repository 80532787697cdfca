#!/bin/bash
# on the GPU box: bench every bio_amd/variants/lib_*.so on a 60 Mb workload
for f in bio_amd/variants/lib_*.so; do
  r=$(BENCH_TUNING_NO_CHECK=1 BIO_AMD_LIB=$PWD/$f timeout -k 10 120 python3 bench.py --length 120000000 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms_avg'], d['roofline']['frac'])")
  echo "$f $r"
done
