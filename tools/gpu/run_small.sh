for l in inc_6e_7x1 k_7x1_768 inc_6e_1x1_768; do
python tools/conv_bench.py --layer $l --which fwd --iters 20 2>&1 | grep -v amdgpu.ids
python tools/conv_bench.py --layer $l --which fwd --iters 20 --const 2>&1 | grep -v amdgpu.ids
done
echo constant payload; timeout 300 tools/probes/probe_stream 2>&1 | grep "^tile" | head -12
echo random payload; PROBE_RANDOM=1 timeout 300 tools/probes/probe_stream 2>&1 | grep "^tile" | head -12
