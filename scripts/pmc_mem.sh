#!/bin/bash
# usage (GPU box): scripts/pmc_mem.sh <tag> <scene> <root> [lib.so ...]   ("default" = in-tree library)
# Memory-pipeline counters of the render kernel (TA / TCP / TCC), one small group per pass.
TAG=$1; SCENE=$2; ROOTN=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd $REPO
LIBS=${@:-default}
for LIB in $LIBS; do
  N=$(basename $LIB .so)
  OUT=$REPO/gpurun_out/pmcmem_${TAG}_$N
  mkdir -p $OUT
  if [ "$LIB" != "default" ]; then export FLUX_HIP_LIB=$REPO/$LIB; else unset FLUX_HIP_LIB; fi
  i=0
  for G in "TA_TA_BUSY_sum TA_BUSY_avr" "TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_READ_sum TCP_GATE_EN1_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    i=$((i+1))
    echo "pass $i: $G"; timeout -k 10 150 rocprofv3 --pmc $G --output-format csv -d $OUT/g$i -- python3 scripts/quick_time.py $SCENE $ROOTN 0 > $OUT/g$i.log 2> $OUT/g$i.err || { echo "pass $i failed: $(tail -1 $OUT/g$i.err)"; }
  done
  grep "rep 1" $OUT/g1.log
  python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "render_" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in agg.items()}
json.dump(a, open(sys.argv[1] + "/summary.json", "w"), indent=1)
for k in sorted(a): print("  %-44s %.6g" % (k, a[k]))
PY
done
