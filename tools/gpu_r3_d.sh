#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_multiview_gpu.py tests/test_densifier.py tests/test_animate_gpu.py tests/test_sds_step_gpu.py tests/test_player_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
unset OMP_NUM_THREADS
for vs in 1 0; do
DWG_VIEW_STREAMS=$vs timeout 600 python bench.py --config c4 --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_c4_vs$vs.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_c4_vs$vs.log") if x.startswith('{"metric"')]
print("c4 view_streams=$vs", (json.loads(l[-1])["value"], json.loads(l[-1])["views_per_s"]) if l else open("$O/bench_c4_vs$vs.log").read()[-2500:])
PY
done
timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_headline.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_headline.log") if x.startswith('{"metric"')]
print("headline", json.loads(l[-1])["value"] if l else open("$O/bench_headline.log").read()[-2500:])
PY
