"""The driver parses the LAST stdout line of bench.py; r03's 20 KB object came back as "parsed": null (BENCH_r03.json).
bench.compact_line() bounds that line.  Checked here on the largest real line the repo holds (r03's, 20 249 bytes) and on a
line inflated well past it; the GPU-side test (tests/test_gpu_sharded.py) asserts the same on what bench.py actually prints."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline")
ROOF_REQUIRED = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_REQUIRED = ("value", "unit", "cores", "kind", "sample")


def full_line():
    with open(os.path.join(ROOT, "profiles", "r03", "bench_default_line.json")) as f:
        return json.load(f)


def check(line):
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    assert "\n" not in text
    back = json.loads(text)
    for k in REQUIRED:
        assert k in back, k
    for k in ROOF_REQUIRED:
        assert k in back["roofline"], k
    for k in CPU_REQUIRED:
        assert k in back["cpu_baseline"], k
    assert "workload" in back["config"] and "model" not in back["config"]
    assert back["roofline"]["bound"] in ("hbm", "mfma")
    assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
    return back


def test_compact_line_of_the_r03_run_fits_and_keeps_the_contract():
    full = full_line()
    assert len(json.dumps(full)) > 16000          # the object that did not parse
    back = check(bench.compact_line(full))
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert back["roofline"]["frac"] == full["roofline"]["frac"]
    assert back["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    s = back["summary"]
    assert s["gcn_ms_per_epoch"] == full["extra"]["gcn_ms_per_epoch"] and s["agnn_ms_per_epoch"] == full["extra"]["agnn_ms_per_epoch"]
    assert s["sddmm_d64_ms"] == full["extra"]["sddmm_d64"]["kernel_ms"]
    assert "sbm_reddit_spmm_ms" in s and "products_d128_sddmm_ms" in s
    assert "datasets" not in back and "extra" not in back


def test_compact_line_stays_bounded_when_the_run_grows():
    full = full_line()
    full["datasets"] = full["datasets"] * 20
    full["extra"]["artifact_shapes"] = full["extra"]["artifact_shapes"] * 50
    full["config"]["workload"] = full["config"]["workload"] * 40
    full["cpu_baseline"]["sample"] = "x" * 5000
    full["roofline"]["kernel"] = "k" * 3000
    for i in range(400):
        full["extra"]["sddmm_d%d" % (1000 + i)] = {"kernel_ms": 1.0, "note": "y" * 100}
    check(bench.compact_line(full))


def test_compact_line_of_a_sharded_run():
    out = {"metric": "m", "value": 1.0, "unit": "GTEPS", "n_gpus": 8, "steps": 5, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f16", "data": "synthetic", "config": {"workload": "w", "parallelism": "p"},
           "roofline": {"bound": "hbm", "kernel": "k", "achieved": 800.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None},
           "extra": {"exchange_in_timed_step": True, "gcn_ms_per_epoch_sharded": 3.0}}
    line = bench.compact_line(out)
    assert len(json.dumps(line)) < 4096
    assert line["summary"]["gcn_ms_per_epoch_sharded"] == 3.0 and line["n_gpus"] == 8


def test_detail_file_holds_everything(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    full = full_line()
    written = bench.write_detail(full)
    assert written == ["bench_detail.json"]
    with open(tmp_path / "bench_detail.json") as f:
        assert json.load(f) == full


def test_plan_only_prints_the_per_rank_memory_plan_of_config_5(tmp_path):
    """`python bench.py --gpus 8 --plan-only`: no GPU, no process group - the per-rank bytes of the
    ogbn-papers100M workload (BASELINE.json configs[4]) against 288 GB, and what one exchange moves."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plan-only"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    doc = json.loads(r.stdout.strip().splitlines()[-1])
    assert doc["plan_only"] and doc["world"] == 8 and doc["nodes"] == 111059956 and doc["edges"] == 3231371744
    assert doc["int32_csr_possible_unsharded"] is False            # why the shard loader exists
    assert len(doc["per_rank"]) == 8 and doc["all_fit"] and 0.05 < doc["max_frac_of_hbm"] < 1.0
    assert sum(x["rows"] for x in doc["per_rank"]) == doc["nodes"] and sum(x["edges"] for x in doc["per_rank"]) == doc["edges"]
    for x in doc["per_rank"]:
        assert x["edges"] < 2 ** 31 and x["rows"] % 16 == 0 or x["rank"] == 7
        assert x["total_bytes"] == sum(x[k] for k in x["parts_summed"]) and x["total_bytes"] < doc["hbm_bytes_per_gpu"]   # (the listed parts add up)
        assert {"csr_bytes", "sgt_metadata_bytes", "plan_bytes_est", "features_bytes", "layer_tensors_bytes"} <= set(x["parts_summed"])
        assert x["exchange_image_ring_bytes"] == 2 * (256 + (x["gathered_rows"] + 1) * 128)     # two 64-column fp16 images: one 128-byte line per row each
        assert x["frac_of_hbm"] <= 0.55 and x["whole_matrix_fp32_exchange"]["frac_of_hbm"] > x["frac_of_hbm"]   # (headroom for config 5)
    ex = doc["exchange_per_spmm"]
    assert ex["fp16_block_bytes"] * 2 == ex["fp32_block_bytes"] and 20 < ex["fp32_ms_link_bound"] < 26    # ~23 ms fp32, ~12 ms fp16
    # with shard files: rows and edges per rank are the files'
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import convert_dataset as C
    import numpy as np
    rng = np.random.default_rng(0)
    s = rng.integers(0, 5000, 60000); d = rng.integers(0, 5000, 60000)
    np.savez(tmp_path / "ei.npz", edge_index=np.stack([s, d]), num_nodes=np.array(5000))
    written = C.convert_sharded(str(tmp_path / "ei.npz"), str(tmp_path / "g"), 4, symmetrize=True, drop_self_loops=True, tmp=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plan-only", "--shard-files", str(tmp_path / "g.rank{rank}of{world}.npz")],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    doc = json.loads(r.stdout.strip().splitlines()[-1])
    assert [x["rows"] for x in doc["per_rank"]] == [w[1] for w in written] and [x["edges"] for x in doc["per_rank"]] == [w[2] for w in written]
