#!/usr/bin/env python3
"""Markdown table of the per-dataset kernel times in a bench line (DESIGN.md section 5 is pasted from this).
usage: tools/bench_table.py profiles/r02/bench_default_line.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ex = d.get("extra", {})
D = d["datasets"][0]["D"]
head = d["datasets"][0]
head.setdefault("spmm_val", ex.get("spmm_agnn_d%d" % D)); head.setdefault("sddmm", ex.get("sddmm_d%d" % D))
head.setdefault("agnn_fused_fwd", ex.get("agnn_fused_fwd_d%d" % D)); head.setdefault("agnn_fused_bwd", ex.get("agnn_fused_bwd_d%d" % D))
def ms(leg): return "-" if not leg else "%.2f" % leg["kernel_ms"]
def fr(leg): return "" if not leg else " (%.3f)" % leg["hbm_frac"]
def tr(leg):
    t = (leg or {}).get("traffic")
    return "-" if not t else "%.2f GB" % (t / 1e9)
print("| workload | SpMM ms (frac) | edge-valued SpMM | SDDMM ms (frac) | fused AGNN fwd / bwd | SpMM HBM-side traffic | SpMM mfma_busy / useful |")
print("|---|---|---|---|---|---|---|")
for x in d["datasets"]:
    if "error" in x: print("| %s | error: %s |" % (x["dataset"], x["error"])); continue
    s = x.get("spmm") or {}
    print("| %s, D = %d | %s%s `%s` | %s | %s%s | %s / %s | %s | %s / %s |" % (
        x["dataset"], x["D"], ms(s), fr(s), s.get("kernel", ""), ms(x.get("spmm_val")), ms(x.get("sddmm")), fr(x.get("sddmm")),
        ms(x.get("agnn_fused_fwd")), ms(x.get("agnn_fused_bwd")), tr(s), s.get("mfma_busy", "-"), s.get("mfma_useful_frac", "-")))
print()
print("headline: %.1f %s, step %.4f ms, kernel %.4f ms (all launches %.4f), frac %.4f, traffic %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms_mean"], d["roofline"]["kernel_ms_mean_all_launches"], d["roofline"]["frac"], d["roofline"]["traffic"]))
print("epochs (headline graph): gcn %.3f ms, agnn %.3f ms; other datasets: %s" % (ex.get("gcn_ms_per_epoch", 0), ex.get("agnn_ms_per_epoch", 0),
      [(x["workload"], x.get("gcn_ms_per_epoch"), x.get("agnn_ms_per_epoch")) for x in d["datasets"] if x.get("agnn_ms_per_epoch") or x.get("gcn_ms_per_epoch")]))
print("cpu_baseline: %s %s on %s threads (%s); torch.sparse %s" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"],
      d["cpu_baseline"]["kind"], d["cpu_baseline"].get("torch_sparse_csr_mm_gteps")))
