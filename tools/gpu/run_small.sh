for e in 1 0; do
echo epi_batch=$e
for l in inc_6e_7x1 inc_5b_5x5 inc_3b_1x1_80 inc_6a_3x3 inc_6e_1x1_768 inc_4a_3x3; do
DIN_CONV_HALO=0 DIN_CONV_EPI_BATCH=$e python tools/conv_bench.py --layer $l --which dgrad --iters 10 2>&1 | grep -v amdgpu.ids | cut -c1-80
done
DIN_CONV_EPI_BATCH=$e timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-180
done
