"""Builds integration/TCGNN_binding.cpp - the reference-side pybind11 binding of INTEGRATION.md section B - as the torch
extension `TCGNN` next to this file (in-tree, so it travels to the GPU box):

    python integration/setup.py build_ext --inplace

A plain CppExtension: no .cu source, so torch's hipify step never runs; it links libtcgnn_hip.so (rpath'd) and torch's own HIP
stream accessor.  Replaces TCGNN_conv/setup.py (CUDAExtension over TCGNN.cpp + TCGNN_kernel.cu, nvcc)."""
import os
import sys

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIBDIR = os.path.join(ROOT, "tc-gnn_atc23_amd", "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

os.chdir(HERE)
setup(
    name="TCGNN",
    ext_modules=[CppExtension(
        name="TCGNN",
        sources=["TCGNN_binding.cpp"],
        include_dirs=[os.path.join(ROOT, "include"), os.path.join(ROCM, "include")],
        define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
        library_dirs=[LIBDIR, os.path.join(ROCM, "lib")],
        libraries=["tcgnn_hip", "c10_hip", "torch_hip", "amdhip64"],
        extra_compile_args=["-O2", "-Wno-deprecated-declarations"],
        extra_link_args=["-Wl,-rpath,$ORIGIN/../tc-gnn_atc23_amd/lib", "-Wl,-rpath," + LIBDIR],
    )],
    cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)},
    script_args=sys.argv[1:] or ["build_ext", "--inplace"],
)
