"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host SGT through the C ABI, symbol
export checks, host-side logic, gloo world_size-2 sharding.  `-m gpu` runs on an MI355X and calls
the HIP kernels through the C ABI (TCGNN module -> libtcgnn_hip.so), checking them against the oracle.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tc-gnn_atc23_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
