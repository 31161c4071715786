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
