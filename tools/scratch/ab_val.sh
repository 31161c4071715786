#!/bin/bash
# rocprofv3 per-kernel averages of forward_AGNN (tools/bench_val.py) with an older library and the tree's, one box
old=$1; gen=$2; out=$GRAFT_REPO_ROOT/gpurun_out/abval; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for which in old new old new; do
  if [ $which = old ]; then export TCGNN_LIB_PATH=$GRAFT_REPO_ROOT/$old; else unset TCGNN_LIB_PATH; fi
  rm -rf $out/raw
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -o p -- python $GRAFT_REPO_ROOT/tools/bench_val.py reddit $gen 64 > $out/log_$which.txt 2>&1
  f=$(find $out/raw -name "*kernel_stats.csv" | head -1)
  echo "== $which $gen"; grep -E "lds_val|val_permute|cold_val" $f | cut -d, -f1-4,6
done
rm -rf $out/raw
