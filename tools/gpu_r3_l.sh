#!/bin/bash
# round 3, pass l: weight-slice prefetch of the small-M GEMMs -- parity + A/B
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3l; mkdir -p $O
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_sd15_full_width_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for np in 0 1 0 1; do
if [ $np = 1 ]; then export DWG_GEMM_NO_PREFETCH=1; else unset DWG_GEMM_NO_PREFETCH; fi
timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_np$np.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_np$np.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("no_prefetch=$np", round(d["value"],2), "steps/s; mfma_all", round(r["mfma_all"]["frac"],4), round(r["mfma_all"]["ms_per_step"],2), "ms; unet_r8/r16/cnet_r8", k.get("conv3x3_unet_r8"), k.get("conv3x3_unet_r16"), k.get("conv3x3_cnet_r8"), "ff_in", k.get("ff_in"), "attn_out", k.get("attn_out"))
    for kk,v in list(r["mfma_kernels"].items())[:4]: print("    %-28s n=%4d avg %.1f us  %.0f TF/s"%(kk, v["launches"], v["avg_launch_ms"]*1e3, v["tflops"]))
else:
    print("no_prefetch=$np FAILED", open("$O/bench_np$np.log").read()[-2500:])
PY
done
