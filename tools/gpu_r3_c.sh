#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
export OMP_NUM_THREADS=16
timeout 900 python -m pytest tests/test_multiview_gpu.py tests/test_densifier.py tests/test_nn_gpu.py tests/test_adam_gpu.py tests/test_sds_step_gpu.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
unset OMP_NUM_THREADS
for i in 1 2; do
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_$i.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/bench_c2_$i.log") if x.startswith('{"metric"')]
print("c2 standalone", json.loads(l[-1])["value"] if l else open("$O/bench_c2_$i.log").read()[-1500:])
PY
done
( time timeout 900 python bench.py ) > $O/bench_default.log 2>&1
echo "bench rc=$?" >> $O/bench_default.log
python - <<PY
import json
l=[x for x in open("$O/bench_default.log") if x.startswith('{"metric"')]
if l:
    d=json.loads(l[-1]); print("headline", d["value"], {k:(v.get("value"), v.get("views_per_s")) for k,v in d["configs"].items()}, "f32", d["by_dtype"]["f32"]["value"])
    print({a:b for a,b in d["kernel_ms_per_step"].items() if a.startswith("gn_")})
else:
    print(open("$O/bench_default.log").read()[-3000:])
PY
tail -5 $O/bench_default.log | cut -c1-300
