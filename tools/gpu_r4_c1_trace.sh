#!/bin/bash
# the kernels of one replayed c1 frame (canonical pose, 10 k Gaussians, 256^2)
mkdir -p gpurun_out/r4s
cd /root/repo
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4s/trace -- python bench.py --config c1 --headline-only --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r4s/c1.log 2>&1
f=$(ls gpurun_out/r4s/trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-1500:]
idx = [i for i, r in enumerate(tail) if "k_preprocess" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
# a frame = from the launch after the previous render to this render: print the launches between two preprocess launches
step = tail[a:b]
t0 = int(step[0]["Start_Timestamp"]); t1 = int(tail[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print("launches per frame: %d; period %.1f us; kernel time %.1f us" % (len(step), (t1 - t0) / 1e3, busy / 1e3))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:100]))
PY
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r4s/c1.log | head -2
