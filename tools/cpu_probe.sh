cd /root/repo
for m in none first spread; do DWG_CPU_PIN=$m python -c "
import bench, json
r = bench.cpu_baseline(100000, 512, backward=True, budget_s=4.0)
print('$m', r.get('s_per_pass'), r.get('pinned'), (r.get('error') or '')[-300:])
"; done
lscpu | grep -i "numa\|socket\|thread\|model name" | head -8
