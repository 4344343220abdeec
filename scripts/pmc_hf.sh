#!/bin/bash
# usage (GPU box): scripts/pmc_hf.sh <tag> <root> lib1.so lib2.so ...   ("default" = in-tree library): SQ counters of the mesh kernel
TAG=$1; ROOTN=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd $REPO
for LIB in "$@"; do
  N=$(basename $LIB .so)
  OUT=$REPO/gpurun_out/pmchf_${TAG}_$N
  mkdir -p $OUT
  if [ "$LIB" != "default" ]; then export FLUX_HIP_LIB=$REPO/$LIB; else unset FLUX_HIP_LIB; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT -- python3 scripts/quick_time.py hf:1000x500 $ROOTN 0 > $OUT/run.log 2> $OUT/run.err || { tail -5 $OUT/run.err; }
  grep "rep 1" $OUT/run.log
  python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "render_" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in agg.items()}
print("%-28s VALU %.4g  ACTIVE_VALU %.4g  SALU %.4g  LDS %.4g  busy %.3f  wait_inst/wave_cycles %.3f" % (
    sys.argv[2], a["SQ_INSTS_VALU"], a["SQ_ACTIVE_INST_VALU"], a["SQ_INSTS_SALU"], a["SQ_INSTS_LDS"],
    a["SQ_ACTIVE_INST_VALU"] * 4 / (a["SQ_BUSY_CYCLES"] / 32 * 1024), a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
PY
done
