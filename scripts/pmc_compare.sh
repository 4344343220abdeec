#!/bin/bash
# usage (GPU box): scripts/pmc_compare.sh <tag> <scene> <root> <variants,comma>
# Two SQ counter passes per render-kernel variant; prints per-kernel averages.  Development aid.
TAG=$1; SCENE=$2; ROOTN=$3; VARS=$4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd $REPO
for PASS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  N=$(echo $PASS | cut -c1-12 | tr ' ' '_')
  OUT=$REPO/gpurun_out/pmcc_${TAG}_$N
  mkdir -p $OUT
  rocprofv3 --pmc $PASS --output-format csv -d $OUT -- python3 scripts/quick_time.py $SCENE $ROOTN $VARS > $OUT/run.log 2> $OUT/run.err || { tail -5 $OUT/run.err; }
done
python3 - "$REPO/gpurun_out" "$TAG" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmcc_" + sys.argv[2] + "_*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "render_" in row["Kernel_Name"]:
            agg[(row["Kernel_Name"].split("(")[0].replace("void flux::", ""), row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-40s %-24s %2d %.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
