#!/bin/bash
# on the GPU box: SQ counters of the pileup kernel for every bio_amd/variants/lib_*.so on one bench config
CFG=${1:-c2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in $R/bio_amd/variants/lib_*.so; do
  n=$(basename $f .so)
  i=0
  for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    rm -rf /tmp/pmc_${n}_$i
    BENCH_TUNING_NO_CHECK=1 BIO_AMD_LIB=$f timeout -s KILL 200 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_${n}_$i -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > /tmp/pmc_${n}_$i.log 2>&1 || { echo "pass failed"; tail -3 /tmp/pmc_${n}_$i.log; exit 1; }
  done
  python3 - $n <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pmc_{n}_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "pileup_kernel" in k and "true>" not in k.split("(")[0][-8:]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(n, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(agg.items())))
PY
done
