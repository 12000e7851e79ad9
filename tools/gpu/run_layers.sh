timeout 600 python bench.py --steps 3 --warmup 3 --no-cpu-baseline --per-layer gpurun_out/per_layer.txt 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --steps 3 --warmup 3 --global-batch 4 --no-cpu-baseline --per-layer gpurun_out/per_layer_b4.txt 2>&1 | tail -1 | cut -c1-200
