#!/bin/bash
# round 3, pass o: compact (LDS-staged) GEMM epilogue -- parity, per-shape times, headline
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3o; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 1200 python -m pytest tests/test_gemm_gpu.py tests/test_nn_gpu.py tests/test_sd15_full_width_gpu.py tests/test_guidance_gpu.py tests/test_sd15_fp16_gpu.py tests/test_sd15_fp32_gpu.py tests/test_animate_gpu.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
unset OMP_NUM_THREADS
PROBE=1 python tools/shape_sweep.py 2>&1 | grep -v amdgpu.ids
python tools/shape_sweep.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_$i.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$i.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("run $i", round(d["value"],2), "steps/s; mfma_all", round(r["mfma_all"]["frac"],4), round(r["mfma_all"]["ms_per_step"],2), "ms; sum kernels", round(sum(k.values()),2))
    for kk,v in list(r["mfma_kernels"].items())[:8]: print("    %-28s n=%4d avg %.1f us  %.0f TF/s"%(kk, v["launches"], v["avg_launch_ms"]*1e3, v["tflops"]))
else:
    print("run $i FAILED", open("$O/bench_$i.log").read()[-2500:])
PY
done
