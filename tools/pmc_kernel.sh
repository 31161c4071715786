#!/usr/bin/env bash
# tools/pmc_kernel.sh <tag> <kernel> [D] [mode]: PMC counters of one kernel on the Reddit shape -> gpurun_out/pmc_<tag>/summary.json
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd); OUT="$ROOT/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_WAVE32_LDS"; do
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
        if "spmm" in k or "sddmm" in k or "agnn_kernel" in k:
            pmc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, d in res.items():
    print(k)
    for c in sorted(d): print("   %-32s %.4g" % (c, d[c]))
PY
find "$OUT" -type f -size +2M -delete
