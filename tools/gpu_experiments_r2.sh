cd /root/repo
for m in copies device; do DWG_GRID_XCD_MODE=$m timeout 150 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_b6_grid_$m.log 2>&1; done
DWG_GN_FOLD=512 timeout 150 python bench.py --steps 10 --warmup 3 > gpurun_out/r2_b6_gnfold.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof6 -o graph -- python /root/repo/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/r2_b6_prof.log 2>&1
cd /root/repo; ls gpurun_out/prof6 | head; find gpurun_out/prof6 -name "*kernel_trace*" -size +20M -delete
grep -o '"value": [0-9.]*, "unit": "SDS' gpurun_out/r2_b6_*.log
