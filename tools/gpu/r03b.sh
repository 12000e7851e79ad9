mkdir -p gpurun_out/r03b; O=gpurun_out/r03b
(time python -m pytest tests -m gpu -q -s 2>&1 | grep -a -v "^Deactivate\|^Hierarchical\|^Activate\|^Dynamic samp" | tail -60) > $O/pytest.log 2>&1
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1
B="python bench.py --no-cpu-baseline --no-extras"
$B --global-batch 4 --steps 20 --warmup 3 --graph off 2>&1 | tail -1 > $O/b4_eager.json
$B --global-batch 4 --steps 20 --warmup 3 --graph on 2>$O/b4_graph.err | tail -1 > $O/b4_graph.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 $B --global-batch 4 --steps 20 --warmup 3 --graph off --force-buckets 2>&1 | grep '"metric"' > $O/b4_eager_buckets.json
MASTER_ADDR=127.0.0.1 MASTER_PORT=29512 $B --global-batch 4 --steps 20 --warmup 3 --graph on --force-buckets 2>$O/b4_graph_buckets.err | grep '"metric"' > $O/b4_graph_buckets.json
$B --steps 10 --warmup 3 --graph off 2>&1 | tail -1 > $O/b32_eager.json
$B --steps 10 --warmup 3 --graph on 2>$O/b32_graph.err | tail -1 > $O/b32_graph.json
DIN_WGRAD_PIPE=0 $B --global-batch 4 --steps 20 --warmup 3 --graph off 2>&1 | tail -1 > $O/b4_eager_nopipe.json
python -m cProfile -o $O/b4.prof bench.py --no-cpu-baseline --no-extras --global-batch 4 --steps 30 --warmup 3 --graph off > /dev/null 2>&1
python -c "
import pstats; p=pstats.Stats('$O/b4.prof'); p.sort_stats('tottime').print_stats(45)" > $O/b4_cprofile.txt 2>&1
DIN_BENCH_TORCH_PROFILE=$O/b4_torch_profile.txt $B --global-batch 4 --steps 3 --warmup 2 --graph off > /dev/null 2>&1
for f in b4_eager b4_graph b4_eager_buckets b4_graph_buckets b32_eager b32_graph b4_eager_nopipe; do echo $f $(python -c "
import json,sys
try:
    d=json.load(open('$O/$f.json')); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['roofline']['frac'], d['config'].get('launch_mode','')[:20])
except Exception as e: print('ERR', e)
"); done
tail -3 $O/pytest.log; tail -2 $O/smoke.log
STRESS_SECONDS=600 tools/stress_gpu_suite.sh run 2 > $O/stress_loop.log 2>&1; tail -8 $O/stress_loop.log
