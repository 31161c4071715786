"""The harness as an artefact: tcgnn_harness mirrors main_tcgnn.py's CLI and - what other tools depend
on - its stdout.  The reference's 1_log2csv.py:10-20 scrapes `dataset=...,` from the printed argparse Namespace and every
line holding "(ms):" except the "Prep." one; 2_tcgnn_single_kernel.py:27-33 drives `--dim H --hidden H --single_kernel`.
The scraper is restated here (scrape) and run over the harness's real output on the GPU; the metadata allocation from the
RAW edge count (main_tcgnn.py:44-47, dataset.py:79) and the device SGT switch are checked in-process."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tc-gnn_atc23_amd", "tcgnn_harness.py")


def scrape(text):
    """1_log2csv.py:10-20, line for line in behaviour: -> (dataset names, time strings)."""
    dataset_li, time_li = [], []
    for line in text.splitlines(keepends=True):
        if "dataset=" in line:
            dataset_li.append(re.findall(r'dataset=.*?,', line)[0].split('=')[1].replace(",", "").replace('\'', ""))
        if "(ms):" in line and "Prep." not in line:
            time_li.append(line.split("(ms):")[1].rstrip("\n").lstrip())
    return dataset_li, time_li


def test_scraper_restatement_reads_the_reference_log_format():
    """The line formats the reference's own logs hold (logs/RTX3090_GCN.log: Namespace line, TC_Blocks / Exp_Edges from C
    stdout, Prep., Train) and the single-kernel line of gnn_conv.py:188."""
    log = ("Namespace(dataset='citeseer', dim=3703, num_layers=2, hidden=16, classes=6, epochs=200, model='gcn', single_kernel=False)\n"
           "TC_Blocks:\t1197\nExp_Edges:\t153216\nPrep. (ms):\t3.025\nTrain (ms):\t 3.031\n"
           "Namespace(dataset='cora', dim=16, num_layers=2, hidden=16, classes=22, epochs=200, model='gcn', single_kernel=True)\n"
           "Prep. (ms):\t2.5\n=> SAG profiling avg (ms): 0.040\n\n")
    assert scrape(log) == (["citeseer", "cora"], ["3.031", "0.040"])


def _run(*argv):
    env = dict(os.environ, PYTHONWARNINGS="ignore")
    r = subprocess.run([sys.executable, HARNESS, *argv], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_single_kernel_run_prints_what_the_reference_scripts_scrape():
    """2_tcgnn_single_kernel.py's command line on the Citeseer-shaped graph (BASELINE.json configs[1])."""
    out = _run("--synthetic", "citeseer", "--dim", "16", "--hidden", "16", "--single_kernel")
    names, times = scrape(out)
    assert names == ["citeseer"] and len(times) == 1 and 0.0 < float(times[0]) < 50.0
    assert re.search(r"^TC_Blocks:\t\d+$", out, re.M) and re.search(r"^Exp_Edges:\t\d+$", out, re.M)          # TCGNN.cpp:225
    assert re.search(r"^Prep\. \(ms\):\t\d+\.\d{3}$", out, re.M)                                                # main_tcgnn.py:54
    assert re.search(r"^=> SAG profiling avg \(ms\): \d+\.\d{3}$", out, re.M)                                  # gnn_conv.py:188
    tc = int(re.search(r"^TC_Blocks:\t(\d+)$", out, re.M).group(1)); ee = int(re.search(r"^Exp_Edges:\t(\d+)$", out, re.M).group(1))
    assert ee == tc * 16 * 8
    assert out.index("TC_Blocks") < out.index("Prep.") < out.index("=> SAG")
    assert "Train (ms)" not in out                                                                             # exit(0) after the profile


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["gcn", "gin", "agnn"])
def test_training_run_prints_the_train_line(model):
    out = _run("--synthetic", "citeseer", "--dim", "32", "--hidden", "16", "--classes", "6", "--epochs", "3", "--model", model)
    names, times = scrape(out)
    assert names == ["citeseer"] and len(times) == 1 and float(times[0]) > 0.0
    assert re.search(r"^Train \(ms\):\t\s*\d+\.\d{3}$", out, re.M)                                             # main_tcgnn.py:181 "{:6.3f}"
    assert out.index("Prep.") < out.index("Train (ms)")


def _toy_npz(path, n=300, m=4000, seed=0):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, size=m); dst = rng.integers(0, n, size=m)
    src, dst = np.concatenate([src, dst, src[:500]]), np.concatenate([dst, src, dst[:500]])   # symmetric, with 500 repeated pairs
    np.savez(path, src_li=src, dst_li=dst, num_nodes=n)
    return len(src)


@pytest.mark.gpu
def test_metadata_is_allocated_from_the_raw_edge_count_and_device_sgt_trains_the_same(tmp_path, capfd):
    """dataset.py:79 counts edges BEFORE duplicates are merged and main_tcgnn.py:45-46 sizes edgeToColumn / edgeToRow by it:
    the arrays are longer than column_index and every entry point must cope.  --gpu_preprocess (device SGT, finished here)
    must give the same training trajectory as the host SGT, bit for bit: same metadata, same kernels, same seeds."""
    sys.path.insert(0, os.path.join(ROOT, "tc-gnn_atc23_amd"))
    import tcgnn_harness as H
    raw = _toy_npz(os.path.join(tmp_path, "toy.npz"))
    base = ["--dataset", "toy", "--graph_dir", str(tmp_path), "--dim", "24", "--hidden", "16", "--classes", "5", "--epochs", "4", "--model", "gcn"]
    r_host = H.run(H.build_parser().parse_args(base), quiet=True)
    assert r_host["num_edges_raw"] == raw and r_host["edge_arrays_len"] == raw and r_host["nnz"] < raw
    assert r_host["num_row_windows"] == (300 + 15) // 16
    r_dev = H.run(H.build_parser().parse_args(base + ["--gpu_preprocess"]), quiet=True)
    assert r_dev["edge_arrays_len"] == raw and r_dev["nnz"] == r_host["nnz"]
    assert np.isfinite(r_host["final_loss"]) and r_dev["final_loss"] == r_host["final_loss"]
    out = capfd.readouterr().out
    assert out.count("TC_Blocks:") == 2                                      # both SGTs report, neither prints anything else when quiet
    blocks = re.findall(r"TC_Blocks:\t(\d+)", out)
    assert blocks[0] == blocks[1]
    # single-kernel profile on the same file, AGNN on it too (SDDMM with padded edge arrays)
    r_sag = H.run(H.build_parser().parse_args(base + ["--single_kernel"]), quiet=True)
    assert r_sag["sag_ms"] > 0
    r_agnn = H.run(H.build_parser().parse_args(base[:-1] + ["agnn"]), quiet=True)
    assert np.isfinite(r_agnn["final_loss"])


@pytest.mark.gpu
def test_reorder_flag_relabels_the_graph_and_leaves_the_scraped_lines_alone():
    """--reorder (tcgnn_graph.community_order before the sparse-graph translation; features and labels follow): the run prints
    its own line in a format the reference's scraper ignores, condenses the shuffled community graph into fewer TC blocks, and
    trains to the same loss as the run on the graph as numbered (same model seed; all-ones labels, permuted features)."""
    base = ("--synthetic", "reddit", "--generator", "sbm_shuffled", "--scale", "0.05", "--dim", "32", "--hidden", "16", "--classes", "6", "--epochs", "3", "--gpu_preprocess")
    plain, reordered = _run(*base), _run(*base, "--reorder")
    assert "Reorder:" in reordered and "Reorder:" not in plain
    assert len(scrape(reordered)[1]) == len(scrape(plain)[1]) == 1
    blocks = lambda text: int(re.findall(r"TC_Blocks:\s*(\d+)", text)[0])
    assert blocks(reordered) < 0.9 * blocks(plain)
