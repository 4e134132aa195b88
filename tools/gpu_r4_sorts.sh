#!/bin/bash
mkdir -p gpurun_out/r4t
cd /root/repo
timeout 900 python -m pytest tests/test_mlp_chain_gpu.py tests/test_animate_gpu.py tests/test_step_graph_gpu.py tests/test_player_gpu.py tests/test_raster_gpu.py tests/test_golden_r2_gpu.py -q > gpurun_out/r4t/test.log 2>&1
grep -n "^E  \|passed\|failed" gpurun_out/r4t/test.log | head -20
for cfg in c1 c2 c5; do
    timeout 300 python bench.py --config $cfg --step-graph --headline-only --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4t/${cfg}.log 2>&1
    echo "$cfg" $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4t/${cfg}.log | head -2)
done
python - <<'PY'
import json
for c in ("c1", "c2", "c5"):
    for l in open('gpurun_out/r4t/%s.log' % c):
        if l.startswith('{"metric"'):
            d = json.loads(l); print(c, {n: v for n, v in d["kernel_ms_per_step"].items() if n.startswith("mlp")})
PY
