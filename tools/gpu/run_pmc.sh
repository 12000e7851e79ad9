cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcb; mkdir -p gpurun_out/pmcb
timeout 700 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcb/fetch -o x -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcb/fetch.log 2>&1; echo "rc=$?"
timeout 700 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcb/write -o x -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmcb/write.log 2>&1; echo "rc=$?"
ls -la gpurun_out/pmcb/*/ | head
