#!/bin/bash
# SQ counters (their own run: no trace domains besides kernel-trace) of the c2 step's kernels, launched eagerly:  bash tools/pmc_sq_c2.sh r04 <commit>
R=${1:-r05}; COMMIT=${2:-unknown}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$R/pmc_sq_c2
mkdir -p $OUT $REPO/profiles
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT -o c2 -- python $REPO/bench.py --config c2 --eager --headline-only --no-cpu-baseline --steps 3 --warmup 2 > $OUT.log 2>&1
cd $REPO
python tools/pmc_sq.py $(find $OUT -name "*counter_collection.csv" | head -1) profiles/${R}_pmc_sq_c2.json $COMMIT | tail -30
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_pmc_sq_c2.json gpurun_out/profiles_$R/
