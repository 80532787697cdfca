#!/bin/bash
# Usage (on the GPU box): [PMC=1] [KERNEL=substr] bash tools/prof_tool.sh <tag> <python tool> [args...]
# Kernel trace + stats of one tool run; with PMC=1 also separate counter passes (never
# combined with a trace domain), averaged per dispatch for kernels whose name holds $KERNEL.
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
tool=$1; shift
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/$tool "$@" > $out/trace.log 2>&1 || { echo trace failed; tail -5 $out/trace.log; exit 1; }
tail -2 $out/trace.log | cut -c1-600
cat $out/trace/*/*kernel_stats.csv | cut -c1-200
[ -n "$PMC" ] || exit 0
i=0
for pmc in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" \
           "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $pmc --output-format csv -d $out/pmc$i -- python3 $R/$tool "$@" > $out/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $out/pmc$i.log; exit 1; }
done
python3 - "$out" "${KERNEL:-kernel}" <<'PY'
import csv, glob, sys, collections
out, want = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if want not in k:
            continue
        agg[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print("== counters (mean per dispatch)", k)
    for c, v in sorted(d.items()):
        print("  %-28s %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
