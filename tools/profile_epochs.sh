#!/usr/bin/env bash
# Per-kernel time of whole training epochs (GCN and AGNN, Reddit shape, hidden 64) under
# rocprofv3 --kernel-trace --stats.  Run ON the GPU box from the repo root:
#   tools/profile_epochs.sh <tag>   -> gpurun_out/epochs_<tag>/{gcn,agnn}_kernel_stats.csv
set -uo pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/epochs_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for model in gcn agnn; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$model" -o ep -- \
    python "$ROOT/tc-gnn_atc23_amd/tcgnn_harness.py" --synthetic reddit --dim 602 --hidden 64 --classes 41 --model $model --epochs 10 --gpu_preprocess \
    > "$OUT/$model.log" 2> "$OUT/$model.err"
  f=$(find "$OUT/$model" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${model}_kernel_stats.csv"
  rm -rf "$OUT/$model"
  tail -3 "$OUT/$model.log"
  head -16 "$OUT/${model}_kernel_stats.csv" | cut -c1-200
done
