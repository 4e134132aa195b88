#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3t; mkdir -p $O
timeout 600 python -m pytest tests/test_nn_gpu.py -m gpu -q -p no:cacheprovider -x -k groupnorm > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
one() {
tag=$1; shift
env "$@" timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_$tag.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$tag.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("%-10s"%"$tag", round(d["value"],2), "steps/s; sum kernels", round(sum(k.values()),2), {a: round(k.get(a,0),3) for a in ("gn_small","gn_stats","gn_finalize","gn_apply")}, "gn total", round(sum(v for a,v in k.items() if a.startswith("gn_")),3))
else:
    print("$tag FAILED", open("$O/bench_$tag.log").read()[-1500:])
PY
}
one deep8 X=1
one max16k DWG_GN_SMALL_MAX=16384
one deep8b X=1
one max16kb DWG_GN_SMALL_MAX=16384
