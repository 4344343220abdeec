#!/bin/bash
# usage: scripts/pmc_quick.sh <tag> "<counters...>" <quick_time args...>   (on the GPU box)
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd $REPO
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- python3 scripts/quick_time.py "$@" > $OUT/run.log 2> $OUT/run.err || { tail -5 $OUT/run.err; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "render_" in row["Kernel_Name"]:
            agg[(row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k[0], k[1], len(v), "%.6g" % (sum(v) / len(v)))
PY
