#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench command.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
case "$TAG" in -*) echo "usage: scripts/profile_gpu.sh <tag> [bench args...] (a tag does not start with -)"; exit 2;; esac
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $REPO
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-rccl-probe $@"
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rccl-probe $@ > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err || { tail -20 $OUT/pmc_fetch.err; exit 1; }
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rccl-probe $@ > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err || { tail -20 $OUT/pmc_write.err; exit 1; }
echo "== pmc SQ"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rccl-probe $@ > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err || { tail -20 $OUT/pmc_sq.err; }
echo "== pmc SQ (2)"
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-rccl-probe $@ > $OUT/bench_pmc_sq2.json 2> $OUT/pmc_sq2.err || { tail -20 $OUT/pmc_sq2.err; }
echo "== FETCH_SIZE calibration (known 1 GiB streams at 16 B/lane and 8 B/lane)"
hipcc --offload-arch=gfx950 -O3 -o $OUT/calib_fetch scripts/calib_fetch.hip 2> $OUT/calib_build.err && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_calib -- $OUT/calib_fetch > $OUT/calib.log 2> $OUT/calib.err || tail -5 $OUT/calib.err
rm -f $OUT/calib_fetch
find $OUT -name '*.csv' | head -30
# keep only small summaries for merging back
find $OUT -name '*kernel_trace.csv' -size +2M -delete
ls -la $OUT/*
