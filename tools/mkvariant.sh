#!/bin/bash
# usage (build container): tools/mkvariant.sh <name> [-DFLAG=..]...   -> falcon-ttdforgnns_amd/lib/libttemb_<name>.so
# A/B variants of the library built with extra -D switches; tools/ab.sh times every lib/libttemb_*.so on the GPU box.
set -e
name=$1; shift
cd "$(dirname "$0")/../falcon-ttdforgnns_amd/csrc"
out=/tmp/ttemb_var/$name
mkdir -p $out
make -s -j4 OUTDIR=$out EXTRA="$*"
cp $out/libttemb_hip.so ../lib/libttemb_$name.so
echo "built lib/libttemb_$name.so ($*)"
