#!/bin/bash
# round 3, pass h: where the GPU idles inside the step (kernel trace -> gap report), graph replay on and off
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
O=$REPO/gpurun_out/r3h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --headline-only --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/graph -o sds -- $B --steps 8 --warmup 3 > $O/graph.log 2>&1
cd $REPO
T=$(find $O/graph -name "*kernel_trace.csv" | head -1)
python tools/gap_report.py $T 6 $O/gaps_graph.json | tee $O/gaps_graph.txt
grep '^{"metric"' $O/graph.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench under trace', d['value'], d['ms_per_step'])"
find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*.db" -delete
