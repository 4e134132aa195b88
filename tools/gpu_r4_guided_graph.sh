#!/bin/bash
# round 4: the guided (c3) step as one captured graph -- parity test, then eager-launch vs whole-step-graph rate
mkdir -p gpurun_out/r4k
cd /root/repo
timeout 900 python -m pytest tests/test_step_graph_gpu.py -x -q -s 2>&1 | tail -15 > gpurun_out/r4k/test.log
timeout 600 python bench.py --headline-only --steps 60 --warmup 10 > gpurun_out/r4k/c3_plans.log 2>&1
timeout 600 python bench.py --headline-only --steps 60 --warmup 10 --step-graph > gpurun_out/r4k/c3_stepgraph.log 2>&1
tail -3 gpurun_out/r4k/test.log
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4k/c3_plans.log | head -2
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4k/c3_stepgraph.log | head -2
tail -5 gpurun_out/r4k/c3_stepgraph.log | cut -c1-600
