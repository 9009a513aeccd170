#!/bin/bash
# usage (on the GPU box): tools/ab2.sh "<kbench args A>" "<kbench args B>" ...  -- every lib/libttemb_*.so x every arg set
cd $GRAFT_REPO_ROOT
for args in "$@"; do
  for lib in falcon-ttdforgnns_amd/lib/libttemb_*.so; do
    echo -n "$(basename $lib .so | sed s/libttemb_//) [$args] : "
    TTEMB_LIB=$PWD/$lib python3 tools/kbench.py $args 2>/dev/null | tail -1
  done
done
