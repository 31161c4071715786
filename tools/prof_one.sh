#!/usr/bin/env bash
# tools/prof_one.sh <tag> <kernel: spmm|spmm_val|sddmm|agnn_fwd|agnn_bwd> [D] [mode]: kernel stats + a few PMC groups of one operator on
# the Reddit shape -> gpurun_out/prof_<tag>/  (run ON the GPU box)
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT="$ROOT/gpurun_out/prof_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python $ROOT/tools/run_kernel_once.py "$@" > /dev/null 2> "$OUT/stats.err"
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"; rm -rf "$OUT/stats"
head -8 "$OUT/kernel_stats.csv" | cut -c1-160
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python $ROOT/tools/run_kernel_once.py "$@" > /dev/null 2> "$OUT/$name.err" || echo "failed: $grp" >> "$OUT/failed.txt"
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0][:70]
        if any(s in k for s in ("spmm", "sddmm", "agnn_kernel")):
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, d in res.items():
    print(k)
    for c in sorted(d): print("   %-28s %.4g" % (c, d[c]))
PY
find "$OUT" -type f -size +2M -delete; find "$OUT" -type d -empty -delete 2>/dev/null
