#!/bin/bash
# round 3, pass j: flash attention -- VGPR-form accumulators (launch bounds 256,2) and the softmax denominator on the MFMA (ones row)
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3j; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 1200 python -m pytest tests/test_nn_gpu.py tests/test_sd15_full_width_gpu.py tests/test_guidance_gpu.py tests/test_sd15_fp16_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
unset OMP_NUM_THREADS
for nl in 0 1 0 1; do
if [ $nl = 1 ]; then export DWG_ATTN_NO_LROW=1; else unset DWG_ATTN_NO_LROW; fi
timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_nl$nl.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_nl$nl.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("no_lrow=$nl", round(d["value"],2), "steps/s; flash d48/d96/d160", k.get("flash_attn_d48"), k.get("flash_attn_d96"), k.get("flash_attn_d160"), "sum kernels", round(sum(k.values()),2))
    for kk,v in r["mfma_kernels"].items():
        if "flash" in kk: print("    %-28s n=%4d avg %.1f us  %.0f TF/s"%(kk, v["launches"], v["avg_launch_ms"]*1e3, v["tflops"]))
else:
    print("no_lrow=$nl FAILED", open("$O/bench_nl$nl.log").read()[-2500:])
PY
done
