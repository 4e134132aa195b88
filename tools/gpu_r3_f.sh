#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3f; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_sd15_full_width_gpu.py tests/test_raster_gpu.py tests/test_baseline_configs_gpu.py tests/test_guidance_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
unset OMP_NUM_THREADS
for w in 1 0; do
DWG_GEMM_WIDE=$w timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_wide$w.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_wide$w.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]
    print("wide=$w", round(d["value"],2), "steps/s; dominant", r["kernel"], round(r["frac"],3), "mfma_all", round(r["mfma_all"]["frac"],3), round(r["mfma_all"]["ms_per_step"],2), "ms")
    for k,v in list(r["mfma_kernels"].items())[:9]: print("    %-28s n=%4d avg %.1f us  %.0f TF/s (%.1f%%)"%(k, v["launches"], v["avg_launch_ms"]*1e3, v["tflops"], 100*v["frac"]))
    print("   raster", r["raster_forward"]["ms"], r["raster_backward"]["ms"], {a:b for a,b in d["kernel_ms_per_step"].items() if a.startswith("splitk")})
else:
    print("wide=$w FAILED", open("$O/bench_wide$w.log").read()[-2500:])
PY
done
