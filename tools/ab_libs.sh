#!/usr/bin/env bash
# A/B of library builds: runs a tool once per libtcgnn_hip*.so variant found in tc-gnn_atc23_amd/lib (TCGNN_LIB_PATH).
# usage (on the GPU box): tools/ab_libs.sh tools/bench_modes.py [grep pattern]
TOOL=$1; PAT=${2:-"D= 64"}
for lib in tc-gnn_atc23_amd/lib/libtcgnn_hip*.so; do
  echo "== $(basename $lib)"
  TCGNN_LIB_PATH=$PWD/$lib timeout 600 python $TOOL 2>&1 | grep -E "$PAT"
done
