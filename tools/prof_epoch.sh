#!/bin/bash
# rocprofv3 kernel stats of one harness training run: tools/prof_epoch.sh <model> <generator> <outfile>
model=$1; gen=$2; out=$3
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_epoch_raw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_epoch_raw -o p -- python $GRAFT_REPO_ROOT/tc-gnn_atc23_amd/tcgnn_harness.py --synthetic reddit --dim 602 --hidden 64 --classes 41 --epochs 10 --gpu_preprocess --generator $gen --model $model > $out.log 2>&1
f=$(find /tmp/prof_epoch_raw -name "*kernel_stats.csv" | head -1)
head -40 $f | cut -d, -f1-6 > $out
rm -rf /tmp/prof_epoch_raw
