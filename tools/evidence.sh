cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02p_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r02p_pytest_gpu.txt
python bench.py --steps 10 --warmup 3 --layers-md gpurun_out/r02p_layers.md > gpurun_out/r02p_bench.json 2> gpurun_out/r02p_bench.err; tail -c 300 gpurun_out/r02p_bench.json
bash tools/prof.sh r02p_prof > gpurun_out/r02p_prof.log 2>&1; DB=$(find gpurun_out/r02p_prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB gpurun_out/r02p_kernel_trace.md > /dev/null 2>&1; head -12 gpurun_out/r02p_kernel_trace.md
bash tools/pmc_hbm.sh > gpurun_out/r02p_pmc.log 2>&1; tail -5 gpurun_out/r02p_pmc.log | cut -c1-300
rm -rf gpurun_out/r02p_prof gpurun_out/pmc_calib_* gpurun_out/pmc_bench_*/pmc* 2>/dev/null
