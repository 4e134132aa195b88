#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_dwg_bind.py tests/test_multiview_gpu.py tests/test_player_gpu.py tests/test_sds_step_gpu.py tests/test_nn_gpu.py tests/test_guidance_gpu.py tests/test_sd15_full_width_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
unset OMP_NUM_THREADS
for mode in shallow deep; do
  if [ $mode = shallow ]; then export DWG_GN_SHALLOW=1; else unset DWG_GN_SHALLOW; fi
  timeout 300 python bench.py --headline-only --no-cpu-baseline > $O/bench_$mode.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/bench_$mode.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); k=d["kernel_ms_per_step"]
    print("$mode", round(d["value"],2), "steps/s", {a:k[a] for a in k if a.startswith("gn_")})
else:
    print("$mode FAILED"); print(open("$O/bench_$mode.log").read()[-2000:])
PY
done
unset DWG_GN_SHALLOW
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/bench_default.log
tail -c 1500 $O/bench_default.log
