#!/usr/bin/env bash
# tools/trace_calls.sh <tag> <op> [D] [mode]: rocprofv3 kernel trace of run_kernel_once.py -> per-launch durations and the gaps between launches of the op's kernel
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT="$ROOT/gpurun_out/trace_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/kt" -o kt -- python $ROOT/tools/run_kernel_once.py "$@" > /dev/null 2> "$OUT/kt.err"
cd "$ROOT"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:50]))
rows.sort()
main = [r for r in rows if any(k in r[2] for k in ("spmm_sync", "sddmm_kernel", "agnn_kernel", "spmm_kernel"))]
if main:
    tail = main[-60:]
    durs = [(e - s) / 1e3 for s, e, _ in tail]
    gaps = [(tail[i + 1][0] - tail[i][1]) / 1e3 for i in range(len(tail) - 1)]
    print("last %d launches of %s: durations us min %.1f mean %.1f max %.1f | gaps us min %.1f median %.1f max %.1f" % (
        len(tail), tail[0][2], min(durs), sum(durs) / len(durs), max(durs), min(gaps), sorted(gaps)[len(gaps) // 2], max(gaps)))
    print("durations:", " ".join("%.0f" % d for d in durs))
    print("gaps:", " ".join("%.1f" % g for g in gaps))
PY
find "$OUT" -type f -size +2M -delete
