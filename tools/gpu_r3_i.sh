#!/bin/bash
# round 3, pass i: split-K reduced inside the GEMM launch (tile semaphores) -- parity + A/B against the two-launch form
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3i; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_sd15_full_width_gpu.py tests/test_guidance_gpu.py tests/test_sd15_fp16_gpu.py tests/test_sds_step_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
unset OMP_NUM_THREADS
for tp in 0 1 0 1; do
if [ $tp = 1 ]; then export DWG_SPLITK_TWO_PASS=1; else unset DWG_SPLITK_TWO_PASS; fi
timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_tp$tp.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_tp$tp.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("two_pass=$tp", round(d["value"],2), "steps/s; mfma_all", round(r["mfma_all"]["frac"],4), round(r["mfma_all"]["ms_per_step"],2), "ms; splitk_epilogue", k.get("splitk_epilogue"), "sum kernels", round(sum(k.values()),2))
else:
    print("two_pass=$tp FAILED", open("$O/bench_tp$tp.log").read()[-2500:])
PY
done
