#!/bin/bash
# usage (on the GPU box): tools/ab_script.sh <python script> [args]  -- runs it once per lib/libttemb_*.so, alternating twice
cd $GRAFT_REPO_ROOT
s=$1; shift
for rep in 1 2; do
  for lib in falcon-ttdforgnns_amd/lib/libttemb_*.so; do
    echo -n "$(basename $lib .so | sed s/libttemb_//) : "
    TTEMB_LIB=$PWD/$lib python3 $s "$@" 2>/dev/null | tail -1
  done
done
