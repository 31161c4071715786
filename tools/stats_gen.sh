#!/usr/bin/env bash
# tools/stats_gen.sh <shape> <generator> <D>: per-kernel durations (rocprofv3 --kernel-trace --stats) of every operator on one synthetic graph
set -uo pipefail
ROOT=$(pwd); OUT="$ROOT/gpurun_out/stats_$1_$2_d$3"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o s -- python $ROOT/tools/run_kernels_for_pmc.py "$1" "$2" "$3" 5 > "$OUT/run.log" 2>&1
f=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"; rm -rf "$OUT/raw"
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("spmm", "sddmm", "agnn_kernel", "convert", "absmax")):
        print("%-90s calls %4s avg %9.1f us" % (n[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
