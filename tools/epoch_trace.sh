#!/usr/bin/env bash
# Kernel-by-kernel timeline of the last training epochs (rocprofv3 --kernel-trace), to see what runs
# between the SpMM launches.  Run ON the GPU box:  tools/epoch_trace.sh <model> <tag>
set -uo pipefail
MODEL=${1:-gcn}; TAG=${2:-r01}
ROOT=$(pwd); OUT="$ROOT/gpurun_out/epochs_$TAG"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_$MODEL" -o ep -- \
  python "$ROOT/tc-gnn_atc23_amd/tcgnn_harness.py" --synthetic reddit --dim 602 --hidden 64 --classes 41 --model $MODEL --epochs 3 --gpu_preprocess > "$OUT/trace_$MODEL.log" 2>&1
python - "$OUT" "$MODEL" <<'PY'
import csv, glob, os, sys
out, model = sys.argv[1:3]
f = glob.glob(os.path.join(out, "trace_" + model, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-260:]
t0 = int(tail[0]["Start_Timestamp"])
with open(os.path.join(out, "timeline_%s.txt" % model), "w") as w:
    prev_end = t0
    for r in tail:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        w.write("%9.1f us  gap %7.1f  dur %8.1f  %s\n" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:110]))
        prev_end = e
PY
rm -rf "$OUT/trace_$MODEL"
