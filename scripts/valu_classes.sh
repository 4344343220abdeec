#!/bin/bash
# usage (GPU box): scripts/valu_classes.sh <tag> <scene> <root> [lib]   -- DYNAMIC VALU instruction classes of the render kernel
# (hardware counters, three rocprofv3 --pmc passes), per 64 samples.  The residual (total - classified) holds what no class counts:
# moves, selects, compares, permutes, lane ops.
TAG=$1; SCENE=$2; ROOTN=$3; LIB=$4
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd $REPO
if [ -n "$LIB" ] && [ "$LIB" != default ]; then export FLUX_HIP_LIB=$REPO/$LIB; fi
OUT=$REPO/gpurun_out/valu_$TAG; mkdir -p $OUT
P1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
P2="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD"
P3="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_WAVES"
k=0
for P in "$P1" "$P2" "$P3"; do
  k=$((k+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/p$k -- python3 scripts/quick_time.py $SCENE $ROOTN 0 > $OUT/run$k.log 2> $OUT/run$k.err || { tail -5 $OUT/run$k.err; exit 1; }
done
grep "rep 1" $OUT/run1.log
python3 - "$OUT" "$SCENE" "$ROOTN" <<'PY'
import csv, glob, sys, collections, json
out, scene, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
agg = collections.defaultdict(list); kname = None
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "render_" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"])); kname = row["Kernel_Name"].split("(")[0]
a = {k: sum(v) / len(v) for k, v in agg.items()}
batches = 800 * 600 * n * n / 64.0
per = {k: v / batches for k, v in a.items()}
classes = ["ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "CVT", "INT32", "INT64"]
tot = per["SQ_INSTS_VALU"]; cl = sum(per.get("SQ_INSTS_VALU_" + c, 0.0) for c in classes)
print("kernel", kname, "| wave-level instructions per 64 samples")
print("  VALU total %.1f" % tot)
for c in classes: print("    %-10s %8.1f  %5.1f %%" % (c, per.get("SQ_INSTS_VALU_" + c, 0.0), 100 * per.get("SQ_INSTS_VALU_" + c, 0.0) / tot))
print("    %-10s %8.1f  %5.1f %%   (moves, selects, compares, permutes, lane ops: counted by no class)" % ("other", tot - cl, 100 * (tot - cl) / tot))
for c in ("SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH"): print("  %-18s %8.1f" % (c[9:], per.get(c, 0.0)))
cyc = a["SQ_BUSY_CYCLES"] / 32.0
print("  VALU busy %.3f  lanes active %.3f  wait_inst/wave_cycles %.3f  wait_lds/wave_cycles %.4f" % (a["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024),
      a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"]), a["SQ_WAIT_INST_ANY"] / a["SQ_WAVE_CYCLES"], a.get("SQ_WAIT_INST_LDS", 0) / a["SQ_WAVE_CYCLES"]))
json.dump({"kernel": kname, "scene": scene, "root": n, "per_64_samples": per, "raw": a}, open(out + "/summary.json", "w"), indent=1)
PY
