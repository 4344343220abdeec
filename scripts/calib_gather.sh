#!/bin/bash
# Runs on the GPU box (via gpurun): what do FETCH_SIZE and the DRAM-side counters report for divergent gathers?
# Usage: scripts/calib_gather.sh <tag>   -> gpurun_out/calib_<tag>/{counters.txt, pmc_<COUNTER>/..., summary.json}
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/calib_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
rocprofv3 -L > $OUT/counters.txt 2> $OUT/counters.err
hipcc --offload-arch=gfx950 -O3 -o $OUT/calib_gather scripts/calib_gather.hip 2> $OUT/build.err || { cat $OUT/build.err; exit 1; }
# one counter (group) per pass; a counter the box does not know fails its own pass only
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_RD_UNCACHED_32B_sum" \
         "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_BUBBLE_sum" "TCC_EA0_RDREQ_IO_CREDIT_STALL_sum" \
         "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" "TCP_TCC_READ_REQ_sum" \
         "TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_EA0_RD_DRAM_sum" "MALL_BANDWIDTH_ALL" "HBM_READ_BYTES" $EXTRA_COUNTERS; do
  D=$OUT/pmc_$(echo $C | tr ' ' '+')
  rocprofv3 --pmc $C --output-format csv -d $D -- $OUT/calib_gather > $D.log 2> $D.err || { echo "pass [$C] failed: $(tail -1 $D.err)"; rm -rf $D; }
done
rm -f $OUT/calib_gather
python3 scripts/calib_gather_summary.py $OUT
