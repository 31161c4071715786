#!/usr/bin/env python3
"""The reference's two committed result tables (logs/profile.csv: single SpMM kernel, D = 16, 200 rounds;
logs/RTX3090_GCN.csv: 2-layer GCN, hidden 16, ms/epoch - both RTX 3090) re-measured on same-size synthetic
graphs through the same harness flow (tcgnn_harness = main_tcgnn.py mirror).  The artifact graphs themselves are
not available, so this compares shapes, not graphs; uniform random graphs condense worse than the real ones.
Prints one CSV line per dataset; bench.py does not call this."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd")): sys.path.insert(0, p)
import tcgnn_harness as H
import tcgnn_graph as G

REF_KERNEL_MS = {"citeseer": 0.040, "cora": 0.066, "pubmed": 0.147, "ppi": 0.537, "PROTEINS_full": 0.115, "OVCAR-8H": 3.199, "Yeast": 2.786,
                 "DD": 0.814, "SW-620H": 3.197, "amazon0505": 3.682, "artist": 1.643, "com-amazon": 1.744, "soc-BlogCatalog": 1.898,
                 "amazon0601": 1.985}   # logs/profile.csv:2-15
REF_EPOCH_MS = {"citeseer": 3.031, "cora": 2.971, "pubmed": 2.793, "ppi": 4.833, "PROTEINS_full": 2.722, "OVCAR-8H": 66.381, "Yeast": 61.057,
                "DD": 11.429, "SW-620H": 68.017, "amazon0505": 23.806, "artist": 4.994, "com-amazon": 17.365, "soc-BlogCatalog": 10.130,
                "amazon0601": 20.310}   # logs/RTX3090_GCN.csv:2-15

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=list(REF_KERNEL_MS))
    ap.add_argument("--epochs", type=int, default=50)
    a = ap.parse_args()
    print("dataset,N,nnz,kernel_ms,ref_rtx3090_kernel_ms,kernel_speedup,gcn_epoch_ms,ref_rtx3090_epoch_ms,epoch_speedup,prep_ms,gcn_epoch_ms_hip_graph")
    for name in a.names:
        n, nnz, dim, classes = G.SHAPES[name]
        base = ["--synthetic", name, "--classes", str(classes), "--gpu_preprocess"]
        k = H.run(H.build_parser().parse_args(base + ["--dim", "16", "--hidden", "16", "--single_kernel"]), quiet=True)
        e = H.run(H.build_parser().parse_args(base + ["--dim", str(dim), "--hidden", "16", "--model", "gcn", "--epochs", str(a.epochs)]), quiet=True)
        try:
            gr = H.run(H.build_parser().parse_args(base + ["--dim", str(dim), "--hidden", "16", "--model", "gcn", "--epochs", str(a.epochs), "--hip_graph"]), quiet=True)["train_ms"]
        except Exception as exc:   # graph capture is an extra; the eager columns stand on their own
            sys.stderr.write("hip_graph leg failed for %s: %s\n" % (name, str(exc)[:300])); gr = float("nan")
        print("%s,%d,%d,%.4f,%.3f,%.2f,%.3f,%.3f,%.2f,%.1f,%.3f" % (name, k["num_nodes"], k["nnz"], k["sag_ms"], REF_KERNEL_MS[name], REF_KERNEL_MS[name] / k["sag_ms"],
                                                                  e["train_ms"], REF_EPOCH_MS[name], REF_EPOCH_MS[name] / e["train_ms"], e["prep_ms"], gr), flush=True)

if __name__ == "__main__":
    main()
