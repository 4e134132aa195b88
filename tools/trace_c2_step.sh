#!/bin/bash
# where the captured c2 step (1.5 ms) goes: per-kernel trace of graph replays + the torch-op inventory of the step
mkdir -p gpurun_out/c2_trace
cd /root/repo
timeout 300 python tools/count_torch_ops.py > gpurun_out/c2_trace/torch_ops.log 2>&1
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c2_trace/trace -- python bench.py --config c2 --step-graph --headline-only --steps 50 --warmup 10 > gpurun_out/c2_trace/c2.log 2>&1
f=$(ls gpurun_out/c2_trace/trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 20 replays: find the period by the k_adam launches
names = [r["Kernel_Name"] for r in rows]
n = len(rows)
tail = rows[-4000:]
# one step = from one 'k_preprocess<' forward to the next
idx = [i for i, r in enumerate(tail) if r["Kernel_Name"].startswith("k_preprocess(") or r["Kernel_Name"].startswith("void k_preprocess") or "k_preprocess" in r["Kernel_Name"] and "bwd" not in r["Kernel_Name"]]
print("preprocess launches in tail:", len(idx))
if len(idx) > 3:
    a, b = idx[-3], idx[-2]
    step = tail[a:b]
    t0 = int(step[0]["Start_Timestamp"]); t1 = int(tail[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    print("launches per step: %d; period %.1f us; kernel time %.1f us" % (len(step), (t1 - t0) / 1e3, busy / 1e3))
    agg = collections.OrderedDict()
    prev_end = None
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = max(prev_end or 0, e)
        print("%8.1f us  +%5.1f gap  %6.1f us  %s" % ((s - t0) / 1e3, gap, (e - s) / 1e3, r["Kernel_Name"][:90]))
PY
