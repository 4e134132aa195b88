#!/bin/bash
# round 3, pass g: the fp16 units (gemm_f16.hip / attention_f16.hip) -- kernel parity, fp16 plans vs oracle / fp32 plans, bench at f16
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3g; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_nn_gpu.py tests/test_sd15_fp16_gpu.py tests/test_sd15_fp32_gpu.py tests/test_guidance_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
unset OMP_NUM_THREADS
cp gpurun_out/parity_fp32.json $O/ 2>/dev/null
for dt in f16 bf16; do
timeout 300 python bench.py --headline-only --no-cpu-baseline --dtype $dt > $O/bench_$dt.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$dt.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]
    print("$dt", round(d["value"],2), "steps/s; dominant", r.get("kernel"), r.get("frac"), "mfma_all", r.get("mfma_all",{}).get("frac"))
else:
    print("$dt FAILED", open("$O/bench_$dt.log").read()[-2500:])
PY
done
