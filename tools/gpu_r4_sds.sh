#!/bin/bash
mkdir -p gpurun_out/r4v
cd /root/repo
timeout 1200 python -m pytest tests/test_sds_kernels_gpu.py tests/test_guidance_gpu.py tests/test_golden_r2_gpu.py tests/test_sds_step_gpu.py tests/test_step_graph_gpu.py tests/test_multiview_gpu.py tests/test_dwg_bind.py tests/test_sd15_fp16_gpu.py -q -m gpu > gpurun_out/r4v/test.log 2>&1
grep -n "^E  \|passed\|failed" gpurun_out/r4v/test.log | head
for mode in 1 0; do
  DWG_SDS_TORCH=$mode timeout 600 python bench.py --headline-only --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r4v/c3_torch$mode.log 2>&1
  echo "sds_torch=$mode" $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4v/c3_torch$mode.log | head -2)
done
timeout 300 python tools/count_torch_ops.py 2>&1 | grep "^== guidance\|^== backward" -A1 | cut -c1-400
