#!/usr/bin/env bash
# tools/pmc_agnn_l2.sh <tag> <mode> [env assignments...] : L2 hit rate and duration of the fused AGNN forward kernel (run_kernel_once.py agnn_fwd 64 <mode>)
set -uo pipefail
TAG=$1; MODE=$2; shift 2
ROOT=$(pwd); OUT="$ROOT/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
env "$@" rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/l2" -o pmc -- python $ROOT/tools/run_kernel_once.py agnn_fwd 64 $MODE > /dev/null 2> "$OUT/l2.err"
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "l2", "**", "*counter_collection*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0][:60]
        if "agnn_kernel" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    h, m = (sum(d[c]) / len(d[c]) for c in ("TCC_HIT_sum", "TCC_MISS_sum"))
    print("%s: L2 hit %.3f (hit %.1f M, miss %.1f M requests)" % (k, h / (h + m), h / 1e6, m / 1e6))
PY
rm -rf "$OUT/l2"
