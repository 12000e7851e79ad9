mkdir -p gpurun_out/exp1
B="python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 5"
for r in 1 2; do
  $B 2>/dev/null | tail -1 > gpurun_out/exp1/base_$r.json
  DIN_WGRAD_PIPE_PAD=21 $B 2>/dev/null | tail -1 > gpurun_out/exp1/pad21_$r.json
done
python tools/conv_bench.py --layer inc_6c_1x7 --which wgrad --iters 3000 | tail -1 > gpurun_out/exp1/w160.log
DIN_WGRAD_PIPE_PAD=21 python tools/conv_bench.py --layer inc_6c_1x7 --which wgrad --iters 3000 | tail -1 >> gpurun_out/exp1/w160.log
