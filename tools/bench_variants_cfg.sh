#!/bin/bash
# on the GPU box: bench every bio_amd/variants/lib_*.so on one bench config (tuning; results may be wrong on purpose)
CFG=${1:-c2}; shift
for f in bio_amd/variants/lib_*.so; do
  r=$(BENCH_TUNING_NO_CHECK=1 BIO_AMD_LIB=$PWD/$f timeout -k 10 200 python3 bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' '.join('%s %s' % (k['ms_avg'], k['frac']) for k in d['roofline']['kernels']), d['ms_per_step'])")
  echo "$CFG $f $r"
done
