#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3e; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_baseline_configs_gpu.py tests/test_player_gpu.py tests/test_animate_gpu.py tests/test_sds_step_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
unset OMP_NUM_THREADS
for c in c5 c2 c3; do
extra=""; [ $c = c3 ] && extra="--headline-only"
timeout 300 python bench.py --config $c --no-cpu-baseline $extra > $O/bench_$c.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_$c.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); k=d["kernel_ms_per_step"]
    print("$c", round(d["value"],2), {a:k[a] for a in k if a.startswith("raster_")})
else:
    print("$c FAILED", open("$O/bench_$c.log").read()[-2000:])
PY
done
