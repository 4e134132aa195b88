#!/bin/bash
# round 3, pass r: split-K / tile-rule knobs re-tuned after the epilogue rewrite (env switches only)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3r; mkdir -p $O
one() {
tag=$1; shift
env "$@" timeout 300 python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$tag.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$tag.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("%-22s"%"$tag", round(d["value"],2), "steps/s; sum kernels", round(sum(k.values()),2), "mfma ms", round(r["mfma_all"]["ms_per_step"],2), "splitk_epi", round(k.get("splitk_epilogue",0),3))
else:
    print("$tag FAILED", open("$O/bench_$tag.log").read()[-800:])
PY
}
one base X=1
one target384 DWG_SPLITK_TARGET=384
one target768 DWG_SPLITK_TARGET=768
one minsteps4 DWG_SPLITK_MINSTEPS=4
one minsteps12 DWG_SPLITK_MINSTEPS=12
one nosplit256 DWG_SPLITK_NOSPLIT=256
one stages2 DWG_GEMM_STAGES=2
one base2 X=1
