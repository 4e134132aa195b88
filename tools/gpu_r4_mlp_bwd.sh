#!/bin/bash
# round 4: fused MLP-chain backward, parallel slab scan, device-side frame tags -- parity, then the c2 step (graph) and the headline
mkdir -p gpurun_out/r4m
cd /root/repo
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_raster_gpu.py tests/test_animate_gpu.py tests/test_step_graph_gpu.py tests/test_player_gpu.py -q -s > gpurun_out/r4m/test.log 2>&1
grep -n "parity\] .*step_graph\|^E  \|passed\|failed" gpurun_out/r4m/test.log | head -20
for mode in 0 1; do
  DWG_MLP_BWD_PER_LAYER=$mode timeout 300 python bench.py --config c2 --step-graph --headline-only --steps 200 --warmup 20 > gpurun_out/r4m/c2_perlayer$mode.log 2>&1
  echo "per_layer=$mode" $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"graph_recaptures": [0-9]*' gpurun_out/r4m/c2_perlayer$mode.log | head -3)
done
timeout 600 python bench.py --headline-only --steps 40 --warmup 8 > gpurun_out/r4m/c3.log 2>&1
echo c3 $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4m/c3.log | head -2)
