#!/bin/bash
# round-3 closing pass: the whole -m gpu suite, then every judged profile regenerated at this code
set -u
cd "$(dirname "$0")/.."
COMMIT=${1:-unknown}
O=gpurun_out/r3final; mkdir -p $O
rm -f gpurun_out/parity_*.json
export OMP_NUM_THREADS=16
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
unset OMP_NUM_THREADS
bash tools/profile_round.sh r03 $COMMIT > $O/profile_round.log 2>&1
tail -30 $O/profile_round.log
for f in parity_fp32 parity_full_width parity_sds_step; do [ -f gpurun_out/$f.json ] && cp gpurun_out/$f.json profiles/r03_$f.json && cp gpurun_out/$f.json gpurun_out/profiles_r03/r03_$f.json; done
ls -la gpurun_out/profiles_r03/
