#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_nn_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
one() {
tag=$1; shift
env "$@" timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_$tag.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$tag.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("$tag", round(d["value"],2), "steps/s; sum kernels", round(sum(k.values()),2), {a: round(k.get(a,0),3) for a in ("gn_stats","gn_finalize","gn_apply","gn_bwd_stats","gn_bwd_apply","splitk_epilogue","layernorm")})
else:
    print("$tag FAILED", open("$O/bench_$tag.log").read()[-1500:])
PY
}
one base X=1
one fold128 DWG_GN_FOLD=128
one small32 DWG_GN_SMALL_CHUNKS=32
one base2 X=1
one fold128b DWG_GN_FOLD=128
