#!/bin/bash
# usage (on the GPU box): tools/ab.sh <kbench args...>   -- runs kbench once per lib/libttemb_*.so, alternating twice
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in falcon-ttdforgnns_amd/lib/libttemb_*.so; do
    echo -n "$(basename $lib) : "
    TTEMB_LIB=$PWD/$lib python3 tools/kbench.py "$@" | tail -1
  done
done
