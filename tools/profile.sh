#!/usr/bin/env bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against.  Run ON the GPU box from
# the repo root:  tools/profile.sh <tag>      -> gpurun_out/profiles_<tag>/
#   1. --kernel-trace --stats of the default bench workload (per-kernel average durations)
#   2. PMC passes, one counter group per run (never combined with tracing domains other than
#      kernel-trace): HBM read bytes (FETCH_SIZE), HBM write bytes (WRITE_SIZE), L2 hit/miss,
#      SQ wave/stall/LDS counters.
set -uo pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT="$ROOT/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --no-extra --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH --steps 20 --warmup 3 > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o pmc -- $BENCH --steps 3 --warmup 1 > /dev/null 2> "$OUT/pmc_$name.err" || echo "pmc group failed: $grp" >> "$OUT/failed.txt"
done
cd "$ROOT"
find "$OUT" -type f | sed "s|$OUT/||" | head -60 > "$OUT/files.txt"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
summary = {}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats*.csv"), recursive=True) + glob.glob(os.path.join(out, "stats*", "**", "*stats*.csv"), recursive=True):
    if "kernel" not in os.path.basename(f):
        continue
    rows = list(csv.DictReader(open(f)))
    summary["kernel_stats"] = [{k: r[k] for k in r} for r in rows[:12]]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        short = k.split("(")[0][:60]
        pmc[short][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
summary["pmc_mean_per_dispatch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items() if "spmm" in k or "sddmm" in k or "convert" in k or "absmax" in k}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1)[:6000])
PY
# keep what is judged (CSV summaries), drop bulky raw traces so gpurun can copy the directory back
find "$OUT" -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.json.gz" -o -size +4M \) -delete
du -sh "$OUT"
