timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b5.json; python - <<'PY'
import json
j=json.loads(open('gpurun_out/b5.json').read()); r=j['roofline']
print(j['value'], j['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['launches'], r['avg_launch_ms'], r.get('sampled_in'), r.get('traffic'), r['all_conv_launches'], j.get('conv_time_frac_sampled_step'))
PY
timeout 600 python bench.py --steps 5 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --steps 3 --warmup 1 --global-batch 4 --no-cpu-baseline --per-layer gpurun_out/pl_test.txt 2>&1 | tail -1 | cut -c1-200; head -3 gpurun_out/pl_test.txt
