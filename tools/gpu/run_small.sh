bash tools/gpu/run_profiles.sh > /dev/null 2>&1
bash tools/gpu/run_layers.sh > /dev/null 2>&1
bash tools/gpu/run_pmc.sh > /dev/null 2>&1
bash tools/gpu/run_prof4.sh > /dev/null 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --forward-only --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_fwd_only.json
timeout 600 python bench.py --steps 5 --warmup 2 --host-images --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_host_images.json
ls gpurun_out/p | head
