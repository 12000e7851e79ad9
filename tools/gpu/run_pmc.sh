cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for H in 1; do
DIN_CONV_HALO=$H timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc/sq_$H -o x -- python tools/conv_bench.py --layer inc_5d_3x3 --which fwd --iters 3 > gpurun_out/pmc/sq_$H.log 2>&1; echo "rc=$?"
DIN_CONV_HALO=$H timeout 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc/sq2_$H -o x -- python tools/conv_bench.py --layer inc_5d_3x3 --which fwd --iters 3 > gpurun_out/pmc/sq2_$H.log 2>&1; echo "rc=$?"
done
