#!/usr/bin/env bash
# Builds the reference's OWN host translation unit (TCGNN_conv/TCGNN.cpp: pybind module with
# `preprocess`, the sparse-graph-translation routine) from the sources where they lie under
# /root/reference, UNMODIFIED, into oracle/_ref/TCGNN_ref*.so.
#
# TEST INFRASTRUCTURE ONLY.  The product never loads this file.
#
# * No reference source is copied and no stand-in source is written: the five CUDA launcher
#   symbols the file declares (TCGNN.cpp:13-52; defined in TCGNN_kernel.cu, which needs nvcc and
#   an NVIDIA GPU and is therefore unbuildable here) are simply left UNDEFINED in the shared
#   object.  ELF function symbols are bound lazily, so `preprocess` (TCGNN.cpp:172-226), which
#   never calls them, runs; calling forward/forward_ef/forward_AGNN on this module would abort
#   with "undefined symbol" - which is the truthful answer in a container without CUDA.
# * The file includes <thrust/sort.h>; rocThrust's headers need HIP mode, hence `-x hip`
#   (host-only compile: --cuda-host-only, no device code is produced).
set -euo pipefail
REF=${TCGNN_REFERENCE_DIR:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/_ref"
SRC="$REF/TCGNN_conv/TCGNN.cpp"
[ -f "$SRC" ] || { echo "build_ref: $SRC not present (GPU box?) - nothing to do"; exit 0; }
mkdir -p "$OUT"
EXT=$(python3 -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
TARGET="$OUT/TCGNN_ref$EXT"
if [ -f "$TARGET" ] && [ "$TARGET" -nt "$SRC" ]; then echo "build_ref: up to date: $TARGET"; exit 0; fi
read -r TORCH_INC TORCH_LIB PY_INC PYBIND_INC CXX11 <<<"$(python3 - <<'PY'
import torch, sysconfig, os, pybind11
from torch.utils import cpp_extension as ce
incs = ce.include_paths()
print(":".join(incs), os.path.join(os.path.dirname(torch.__file__), "lib"),
      sysconfig.get_paths()["include"], pybind11.get_include(),
      int(torch._C._GLIBCXX_USE_CXX11_ABI))
PY
)"
INCFLAGS=""
IFS=':' read -ra P <<<"$TORCH_INC"; for p in "${P[@]}"; do INCFLAGS="$INCFLAGS -isystem $p"; done
set -x
/opt/rocm/bin/hipcc -x hip --cuda-host-only -O2 -std=c++17 -fPIC -shared \
  -Wno-deprecated-declarations -Wno-unused-result -w \
  -DTORCH_EXTENSION_NAME=TCGNN_ref -DTORCH_API_INCLUDE_EXTENSION_H \
  -D_GLIBCXX_USE_CXX11_ABI=$CXX11 -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 \
  $INCFLAGS -isystem "$PY_INC" -isystem "$PYBIND_INC" -isystem /opt/rocm/include \
  "$SRC" -o "$TARGET" \
  -L"$TORCH_LIB" -ltorch -ltorch_cpu -lc10 -ltorch_python -Wl,-rpath,"$TORCH_LIB" \
  -Wl,--allow-shlib-undefined -Wl,-z,lazy
set +x
echo "build_ref: built $TARGET"
