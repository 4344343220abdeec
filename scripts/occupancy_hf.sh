#!/bin/bash
# usage (GPU box): scripts/occupancy_hf.sh <tag> <scene> <root> lib1.so ...  ("default" = in-tree library)
# Resident waves per SIMD of the render kernel, from SQ_WAVE_CYCLES / SQ_BUSY_CYCLES (calibrated by a build whose occupancy is known).
TAG=$1; SCENE=$2; ROOTN=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd $REPO
for LIB in "$@"; do
  N=$(basename $LIB .so)
  OUT=$REPO/gpurun_out/occ_${TAG}_$N
  mkdir -p $OUT
  if [ "$LIB" != "default" ]; then export FLUX_HIP_LIB=$REPO/$LIB; else unset FLUX_HIP_LIB; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LEVEL_WAVES --output-format csv -d $OUT -- python3 scripts/quick_time.py $SCENE $ROOTN 0 > $OUT/run.log 2> $OUT/run.err || { tail -5 $OUT/run.err; }
  grep "rep 1" $OUT/run.log
  python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
meta = {}
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "render_" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta = {k: row[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size") if k in row}
a = {k: sum(v) / len(v) for k, v in agg.items()}
cyc = a["SQ_BUSY_CYCLES"] / 32.0
print("%-24s %s" % (sys.argv[2], meta))
print("   waves %.4g  wave_cycles/(busy_cycles/32 * 1024 SIMDs) = %.3f   level_waves/busy = %s   VALU busy %.3f  wait_inst/wave_cycles %.3f" % (
    a["SQ_WAVES"], a["SQ_WAVE_CYCLES"] / (cyc * 1024.0), ("%.3f" % (a["SQ_LEVEL_WAVES"] / cyc / 1024.0)) if "SQ_LEVEL_WAVES" in a else "n/a",
    a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024), a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"]))
PY
done
