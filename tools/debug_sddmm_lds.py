#!/usr/bin/env python3
"""Debug aid: LDS-resident SDDMM against the gather walk on one graph, with the workspace poisoned (NaN) so that stream positions
nobody wrote show up; prints where the two differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tc-gnn_atc23_amd"), os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np, torch
import TCGNN, tcgnn_capi as c, graphs
dev = torch.device("cuda:0")
n, deg, D = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
rp, col = graphs.uniform_graph(n, deg, seed=21)
bp, e2c, e2r, _ = graphs.host_sgt(rp, col)
meta = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (rp, col, bp, e2c, e2r)]
X = torch.randn(n, D, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
if os.environ.get("ONES"): X = torch.ones(n, D, device=dev)
c.lib.tcgnn_set_spmm_mode(1); ref = TCGNN.forward_ef(X, *meta)[0]
c.lib.tcgnn_set_spmm_mode(3)
plan = TCGNN._plan_for(*meta)
need = c.lib.tcgnn_workspace_bytes(plan, D)
for trial in range(3):
    ws = torch.full((need + 512,), 0xFF, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    ef = torch.full((len(col),), float("nan"), device=dev)
    c.check(c.lib.tcgnn_sddmm(plan, X.data_ptr(), ef.data_ptr(), D, ws.data_ptr() + off, ws.numel() - off, torch.cuda.current_stream().cuda_stream), "sddmm")
    torch.cuda.synchronize()
    if os.environ.get("TCGNN_SD_DBG") == "2":
        perm = ef.cpu().numpy().astype(np.int64)
        win0 = rp[(np.searchsorted(rp, np.arange(len(col)), side="right") - 1) // 16 * 16]
        nwin = (rp[np.minimum((np.searchsorted(rp, np.arange(len(col)), side="right") - 1) // 16 * 16 + 16, n)] - win0)
        print("trial", trial, "perm >= window size:", int((perm >= nwin).sum()), " distinct per window ok:", len(np.unique(win0 + perm)) == len(col), "max", perm.max())
        np.save("/tmp/perm.npy", perm)
        print("   first 40:", perm[:40].tolist()); print("   hist:", np.bincount(np.minimum(perm, 63))[:40].tolist()); print("   raw ef[:8]", ef[:8].cpu().numpy())
        continue
    if os.environ.get("TCGNN_SD_DBG") == "1":
        if os.path.exists("/tmp/perm.npy") and not os.environ.get("ONES"):
            perm = np.load("/tmp/perm.npy")
            win0 = rp[(np.searchsorted(rp, np.arange(len(col)), side="right") - 1) // 16 * 16]
            emul = ef.cpu().numpy()[win0 + perm]
            r = ref.cpu().numpy()
            badm = ~(np.abs(emul - r) <= 1e-3 * (1 + np.abs(r)))
            print("trial", trial, "host emulation of pass 2 from the device stream and device perm: bad", int(badm.sum()))
            continue
        z = torch.nonzero(ef != float(D)).flatten().cpu().numpy()
        print("trial", trial, "stream positions != D:", len(z), "values", ef[torch.from_numpy(z[:10]).to(dev)].cpu().numpy() if len(z) else "")
        if len(z):
            w = np.searchsorted(rp[::16], z, side="right") - 1
            import collections
            print("   windows hit:", len(set(w.tolist())), " first positions (window, offset in window):", [(int(a), int(b - rp[16 * a])) for a, b in zip(w[:24], z[:24])])
            runs = np.split(z, np.where(np.diff(z) != 1)[0] + 1)
            print("   run lengths:", collections.Counter(len(r) for r in runs).most_common(8))
        continue
    bad = ~((ef - ref).abs() <= 1e-3 * (1 + ref.abs()))
    print("trial", trial, "kernel", TCGNN.last_kernel(*meta), "edges", len(col), "bad", int(bad.sum()), "nan", int(torch.isnan(ef).sum()))
    if bad.any():
        idx = torch.nonzero(bad).flatten()[:20].cpu().numpy()
        rows = np.searchsorted(rp, idx, side="right") - 1
        for e, r in zip(idx, rows):
            print("   e=%d row=%d (win %d, i=%d) col=%d range=%d local=%d got=%s ref=%.4f" % (e, r, r // 16, r % 16, col[e], col[e] // 760, col[e] % 760, float(ef[e]), float(ref[e])))
        import collections
        allbad = torch.nonzero(bad).flatten().cpu().numpy()
        print("   bad by local row // 32:", sorted(collections.Counter((col[allbad] % 760) // 32).items()))
        print("   bad by range:", sorted(collections.Counter(col[allbad] // 760).items()))
        rws = np.searchsorted(rp, allbad, side="right") - 1
        print("   bad by window %% 16:", sorted(collections.Counter((rws // 16) % 16).items()))
        print("   bad by row i:", sorted(collections.Counter(rws % 16).items()))
        print("   bad values:", collections.Counter(np.round(ef[torch.from_numpy(allbad).to(dev)].cpu().numpy(), 3)).most_common(5))
c.lib.tcgnn_set_spmm_mode(0)
