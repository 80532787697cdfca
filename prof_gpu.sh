#!/bin/bash
# Usage (on the GPU box): bash prof_gpu.sh <tag> [bench args...]
# Kernel trace + stats, then separate PMC passes (never combined with other trace domains).
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline $@"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py $ARGS > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- python3 $R/bench.py $ARGS > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc$i.log; exit 1; }
done
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    print(open(f).read())
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "pileup_kernel" not in k and "plan_windows" not in k:
            continue
        agg[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print("== counters (mean per dispatch)", k)
    for c, v in sorted(d.items()):
        print("  %-24s %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
