#!/bin/bash
# Regenerates the judged profiles of a round at HEAD on a GPU box:  bash tools/profile_round.sh r03 [commit]
# (the box has no .git: pass `git rev-parse --short HEAD` as the second argument; the PMC profile is also stamped with a hash of the kernel sources)
# (kernel-trace / stats in their own runs, PMC counters in their own runs -- gpurun refuses the combination)
set -u
R=${1:-r05}
COMMIT=${2:-unknown}
# DWG_PROFILE_PARTS: which passes to run (default all): eager graph pmc bench
PARTS=${DWG_PROFILE_PARTS:-"eager graph pmc sq bench small calib parity"}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$R
mkdir -p $OUT $REPO/profiles
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --headline-only --no-cpu-baseline"
has eager && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eager -o sds -- $B --steps 5 --warmup 2 --eager > $OUT/eager.log 2>&1
has graph && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/graph -o sds -- $B --steps 5 --warmup 2 > $OUT/graph.log 2>&1
has pmc && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o sds -- $B --steps 2 --warmup 1 --eager > $OUT/pmc_fetch.log 2>&1
has pmc && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o sds -- $B --steps 2 --warmup 1 --eager > $OUT/pmc_write.log 2>&1
has sq && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq -o sds -- $B --steps 2 --warmup 1 --eager > $OUT/pmc_sq.log 2>&1
cd $REPO
has sq && python tools/pmc_sq.py $(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1) profiles/${R}_pmc_sq.json $COMMIT > $OUT/pmc_sq_summary.log 2>&1
has sq && tail -18 $OUT/pmc_sq_summary.log
find $OUT -name "*kernel_stats.csv" | head
# the judged summaries are the STEADY STATE (tools/steady_stats.py: the trace from the first training step's first kernel on -- no weight
# packing, no one-off library launches); rocprofv3's own whole-process summaries are kept next to them as *_whole_process_*
has eager && python tools/steady_stats.py $OUT/eager profiles/${R}_sds_step_eager_kernel_stats.csv
has graph && python tools/steady_stats.py $OUT/graph profiles/${R}_sds_step_graph_kernel_stats.csv
has eager && cp $(find $OUT/eager -name "*kernel_stats.csv" | head -1) profiles/${R}_sds_step_eager_whole_process_kernel_stats.csv
has graph && cp $(find $OUT/graph -name "*kernel_stats.csv" | head -1) profiles/${R}_sds_step_graph_whole_process_kernel_stats.csv
has pmc && python tools/pmc_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $OUT/pmc_write -name "*counter_collection.csv" | head -1) profiles/${R}_pmc_traffic.json $COMMIT ${DWG_PROFILE_DTYPE:-f32x} > $OUT/pmc_traffic.log 2>&1
has pmc && tail -2 $OUT/pmc_traffic.log
has eager && grep '^{"metric"' $OUT/eager.log | tail -1 > profiles/${R}_sds_step_eager_bench_line.json
# the two captured small configurations: per-kernel statistics of the replayed c2 step and c1 frame
if has small; then
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o c2 -- python $REPO/bench.py --config c2 --step-graph --headline-only --no-cpu-baseline --steps 200 --warmup 20 > $OUT/c2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c1 -o c1 -- python $REPO/bench.py --config c1 --headline-only --no-cpu-baseline --steps 200 --warmup 20 > $OUT/c1.log 2>&1
cd $REPO
python tools/steady_stats.py $OUT/c2 profiles/${R}_c2_step_graph_kernel_stats.csv
python tools/steady_stats.py $OUT/c1 profiles/${R}_c1_frame_graph_kernel_stats.csv
# c5 (the animation frame: where the rasterizer IS the frame)
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o c5 -- python $REPO/bench.py --config c5 --headline-only --no-cpu-baseline --steps 100 --warmup 10 > $OUT/c5.log 2>&1
cd $REPO
python tools/steady_stats.py $OUT/c5 profiles/${R}_c5_frame_kernel_stats.csv
fi
# bench lines (the default command, then the two other BASELINE configs)
if has bench; then
# the default command carries c1 / c2 / c4 (8 views on one GPU) / c5 and the fp32 line as attachments
T0=$(date +%s); timeout 900 python bench.py > $OUT/bench_default.log 2>&1; echo "default bench.py run: $(( $(date +%s) - T0 )) s" | tee $OUT/bench_default.time; grep '^{"metric"' $OUT/bench_default.log | tail -1 > profiles/${R}_bench_line.json
fi
has calib && bash tools/calib_fetch.sh $R > $OUT/calib.log 2>&1
has calib && tail -20 $OUT/calib.log
# parity records at HEAD (round-5 verdict, weak #3): the parity tests write per-case max / q99.9 / threshold-flip counts / gradient rel-L2 to
# gpurun_out/parity_*.json; the round's copies are what profiles/ keeps
if has parity; then
rm -f gpurun_out/parity_*.json
timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_baseline_configs_gpu.py tests/test_sd15_f32x_gpu.py tests/test_sd15_fp32_gpu.py tests/test_sd15_fp16_gpu.py tests/test_sd15_full_width_gpu.py tests/test_golden_r2_gpu.py tests/test_sds_step_gpu.py tests/test_guidance_gpu.py -q -m gpu > $OUT/parity_tests.log 2>&1
tail -3 $OUT/parity_tests.log
for f in gpurun_out/parity_*.json; do [ -f "$f" ] && cp "$f" profiles/${R}_$(basename $f); done
fi
# large raw traces stay on the box
find $OUT -name "*kernel_trace.csv" -size +8M -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*.db" -delete
mkdir -p gpurun_out/profiles_$R && cp profiles/${R}_* gpurun_out/profiles_$R/
ls -la profiles | tail -12
