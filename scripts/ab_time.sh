#!/bin/bash
# usage (GPU box): scripts/ab_time.sh <scene> <root> <variant> lib1 lib2 ...   ("default" = the in-tree library; others are
# names under flux_amd/variants/libflux_hip_<name>.so): kernel ms of the second of two frames, same box, same process order
SCENE=$1; ROOTN=$2; VAR=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
for lib in "$@"; do
  if [ "$lib" = default ]; then unset FLUX_HIP_LIB; else export FLUX_HIP_LIB=$REPO/flux_amd/variants/libflux_hip_$lib.so; fi
  echo "$SCENE n=$ROOTN $lib: $(python3 scripts/quick_time.py $SCENE $ROOTN $VAR | grep 'rep 1' | cut -c1-100)"
done
