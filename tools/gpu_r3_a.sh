#!/bin/bash
# round 3, first GPU pass: the whole -m gpu suite (with durations), then the default bench line
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r3a
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests -m gpu -q --durations=25 -p no:cacheprovider > gpurun_out/r3a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -60 gpurun_out/r3a/pytest.log
unset OMP_NUM_THREADS
( time timeout 900 python bench.py ) > gpurun_out/r3a/bench_default.log 2>&1
echo "bench rc=$?" >> gpurun_out/r3a/bench_default.log
tail -c 6000 gpurun_out/r3a/bench_default.log
