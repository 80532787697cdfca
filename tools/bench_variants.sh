#!/bin/bash
# on the GPU box: bench every bio_amd/variants/lib_*.so on the bench workload (LEN positions, default chr1)
LEN=${LEN:-248956422}
for f in bio_amd/variants/lib_*.so; do
  r=$(BENCH_TUNING_NO_CHECK=1 BIO_AMD_LIB=$PWD/$f timeout -k 10 200 python3 bench.py --length $LEN --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['kernel_ms_avg'], d['roofline']['frac'])")
  echo "$f $r"
done
