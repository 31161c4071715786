#!/bin/bash
# rocprofv3 kernel stats of forward_AGNN on one generator, with and without dense entries (TCGNN_LDS_DENSE_COLS)
#   tools/prof_val.sh <generator> <outdir>
gen=$1; out=$2; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for dense in default off; do
  if [ $dense = off ]; then export TCGNN_LDS_DENSE_COLS=1000000000; else unset TCGNN_LDS_DENSE_COLS; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_$dense -o p -- python $GRAFT_REPO_ROOT/tools/bench_val.py reddit $gen 64 > $out/val_${gen}_$dense.log 2>&1
  f=$(find $out/raw_$dense -name "*kernel_stats.csv" | head -1)
  grep -E "val|cold" $f | cut -d, -f1-6 > $out/val_${gen}_${dense}_stats.txt
  rm -rf $out/raw_$dense
done
