#!/bin/bash
# A/B of two builds of libtcgnn_hip.so on ONE box (boxes differ by 10 % and more): tools/lds_dbg.py on the named generators with the
# library under tools/bin/ (an older build, TCGNN_LIB_PATH) and with the tree's own, alternating.   tools/ab_lds.sh OLD.so gen [gen ...]
old=$1; shift
for g in "$@"; do
  for rep in 1 2; do
    echo "== $g old"; GEN=$g TCGNN_LIB_PATH=$old python tools/lds_dbg.py 64 2>&1 | grep "dbg="
    echo "== $g new"; GEN=$g python tools/lds_dbg.py 64 2>&1 | grep "dbg="
  done
done
