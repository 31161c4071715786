#!/usr/bin/env bash
# tools/pmc_quick.sh <tag> [env assignments...] : L2 hit/miss, fabric read bytes and wave-state counters
# of the bench's SpMM kernel under the given environment (e.g. TCGNN_SPMM_MODE=2), summarised as JSON.
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT="$ROOT/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum TA_TA_BUSY_sum"; do
  name=$(echo "$grp" | tr ' ' '_' | cut -c1-30)
  env "$@" rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$name" -o pmc -- python $ROOT/bench.py --no-extra --no-cpu --steps 3 --warmup 1 > /dev/null 2> "$OUT/$name.err" || echo "failed: $grp" >> "$OUT/failed.txt"
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection*.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "").split("(")[0][:70]
        if "spmm" in k or "sddmm" in k:
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
for k, d in res.items():
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "FETCH_SIZE" in d:
        d["fabric_read_GB_corrected_x2"] = d["FETCH_SIZE"] * 2 * 1024 / 1e9
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find "$OUT" -type f -size +2M -delete
