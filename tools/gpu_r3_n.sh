#!/bin/bash
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3n
run() { echo "== $*"; env "$@" python tools/shape_sweep.py 2>&1 | grep -v amdgpu.ids; }
run X=1
run DWG_GEMM_DEBUG=1
run DWG_GEMM_DEBUG=2
run DWG_SPLITK_NOSPLIT=1
