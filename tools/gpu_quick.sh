#!/bin/bash
# quick check: GEMM / block parity tests + two headline bench runs with the per-label kernel times that matter
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/quick; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_sd15_full_width_gpu.py ${EXTRA_TESTS:-} -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
unset OMP_NUM_THREADS
for i in 1 2; do
timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_$i.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$i.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); r=d["roofline"]; k=d["kernel_ms_per_step"]
    print("run $i", round(d["value"],2), "steps/s; mfma_all", round(r["mfma_all"]["frac"],4), round(r["mfma_all"]["ms_per_step"],2), "ms; sum kernels", round(sum(k.values()),2))
    print("   ", {a: round(b,3) for a,b in sorted(k.items(), key=lambda x:-x[1])[:16]})
else:
    print("run $i FAILED", open("$O/bench_$i.log").read()[-2500:])
PY
done
