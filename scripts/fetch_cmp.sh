#!/bin/bash
# usage: fetch_cmp.sh  -> FETCH_SIZE per variant at n=128
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in ${VARIANTS:-rowmajor grouped}; do
  export FLUX_HIP_LIB=$GRAFT_REPO_ROOT/flux_amd/variants/libflux_hip_$v.so
  OUT=gpurun_out/pmcq_fetch_$v; mkdir -p $OUT
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 scripts/quick_time.py demo2 128 2 > $OUT/run.log 2> $OUT/run.err || tail -3 $OUT/run.err
  grep "rep 1" $OUT/run.log
  python3 - $OUT <<'PY'
import csv, glob, sys
v=[float(r["Counter_Value"]) for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv") for r in csv.DictReader(open(f)) if "render_refill" in r["Kernel_Name"]]
print("FETCH_SIZE KB avg", sum(v)/len(v), "-> x2 GB", sum(v)/len(v)*2*1024/1e9)
PY
done
