#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3q; mkdir -p $O
one() {
tag=$1; shift
env "$@" timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_$tag.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$tag.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("$tag", round(d["value"],2), "steps/s; sum kernels", round(sum(k.values()),2), "mfma ms", round(r["mfma_all"]["ms_per_step"],2), {a: round(k.get(a,0),3) for a in ("gn_stats","gn_finalize","gn_finalize_cols","gn_apply","splitk_epilogue")})
else:
    print("$tag FAILED", open("$O/bench_$tag.log").read()[-1500:])
PY
}
one nofold DWG_GN_COLFOLD=0
one nocolstats DWG_GN_NO_COLSTATS=1
one nofold2 DWG_GN_COLFOLD=0
one nocolstats2 DWG_GN_NO_COLSTATS=1
