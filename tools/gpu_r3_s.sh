#!/bin/bash
# round 3, pass s: one-launch GroupNorm for the small latent levels -- parity + A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3s; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_sd15_full_width_gpu.py tests/test_sd15_fp32_gpu.py tests/test_sd15_fp16_gpu.py tests/test_guidance_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
unset OMP_NUM_THREADS
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
one small X=1
one nosmall DWG_GN_NO_SMALL=1
one small2 X=1
one nosmall2 DWG_GN_NO_SMALL=1
