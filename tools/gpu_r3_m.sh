#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3m; mkdir -p $O
run() { echo "== $*"; env "$@" python tools/shape_sweep.py 2>&1 | grep -v amdgpu.ids; }
run X=1
run DWG_SPLITK_TARGET=256
run DWG_SPLITK_TARGET=1024 DWG_SPLITK_MINSTEPS=4
run DWG_SPLITK_NOSPLIT=1
run DWG_GEMM_STAGES=2
run DWG_GEMM_NO_NARROW=1
run DWG_CONV_NO_FAST=1 SHAPES=conv
run DWG_CONV_PATCH_MINM=128 SHAPES=conv3x3
