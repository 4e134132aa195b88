#!/bin/bash
mkdir -p gpurun_out/r4p
cd /root/repo
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_animate_gpu.py tests/test_step_graph_gpu.py tests/test_sds_step_gpu.py tests/test_multiview_gpu.py tests/test_golden_r2_gpu.py -q > gpurun_out/r4p/test.log 2>&1
grep -n "^E  \|passed\|failed" gpurun_out/r4p/test.log | head -20
timeout 300 python bench.py --config c2 --step-graph --headline-only --steps 200 --warmup 20 > gpurun_out/r4p/c2.log 2>&1
echo c2 $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4p/c2.log | head -2) $(grep -o '"graph_recaptures": [0-9]*' gpurun_out/r4p/c2.log)
timeout 600 python bench.py --headline-only --steps 40 --warmup 8 > gpurun_out/r4p/c3.log 2>&1
echo c3 $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4p/c3.log | head -2)
