#!/usr/bin/env bash
# tools/pmc_calls.sh <tag> <op> [D] [mode]: PMC counters of one operator call, SUMMED over the launches of a call (the slice-synchronised
# walk is one launch per slice round) -> gpurun_out/pmc_<tag>/summary.txt.  Shape / generator: TCGNN_PROFILE_SHAPE / TCGNN_PROFILE_GEN.
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT="$ROOT/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python $ROOT/tools/run_kernel_once.py "$@" > /dev/null 2> "$OUT/$name.err" || echo "failed: $grp" >> "$OUT/failed.txt"
done
cd "$ROOT"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, sys, collections
out = sys.argv[1]
pmc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0][:60]
        if "spmm" in k or "sddmm" in k or "agnn_kernel" in k:
            pmc[k][r["Counter_Name"]] += float(r["Counter_Value"])
CALLS = 5.0   # run_kernel_once.py: one warm call (the fused pair) or none + 4: close enough for ratios; bytes below are per call assuming 4 calls of the op itself
for k, d in pmc.items():
    calls = 4.0
    print(k)
    for c in sorted(d): print("   %-24s %.4g per call" % (c, d[c] / calls))
    if "TCC_HIT_sum" in d: print("   L2 hit rate %.3f" % (d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])))
    if "FETCH_SIZE" in d: print("   fabric bytes per call: reads %.2f GB (2 x FETCH_SIZE x 1024) + writes %.2f GB" % (2 * d["FETCH_SIZE"] * 1024 / calls / 1e9, d.get("WRITE_SIZE", 0) * 1024 / calls / 1e9))
PY
find "$OUT" -type f -size +2M -delete
