#!/bin/bash
mkdir -p gpurun_out/r4u
cd /root/repo
timeout 900 python -m pytest tests/test_gemm_gpu.py tests/test_gemm_x_gpu.py tests/test_sd15_f32x_gpu.py -q > gpurun_out/r4u/test.log 2>&1
grep -n "^E  \|passed\|failed" gpurun_out/r4u/test.log | head
timeout 600 python bench.py --headline-only --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r4u/c3.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r4u/c3.log'):
    if l.startswith('{"metric"'):
        d = json.loads(l); k = d["kernel_ms_per_step"]
        print(d["value"], d["ms_per_step"], "splitk", k.get("splitk_epilogue"), "sum", sum(k.values()))
PY
