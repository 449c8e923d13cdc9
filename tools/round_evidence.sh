cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r04_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r04_pytest_gpu.txt
bash tools/evidence.sh r04 notests 2>&1 | tail -15
bash tools/pmc_split_gemm.sh r04 > gpurun_out/r04_pmc_sg.log 2>&1; tail -3 gpurun_out/r04_pmc_sg.log
bash tools/pmc_wino2.sh r04 > gpurun_out/r04_pmc_w2.log 2>&1; tail -3 gpurun_out/r04_pmc_w2.log
ROUND=r04 bash tools/pmc_hbm.sh > gpurun_out/r04_pmc_hbm.log 2>&1; tail -3 gpurun_out/r04_pmc_hbm.log
FCD_LIB=build_exp/libfcdgan_w2time.so python tools/w2_segments.py --md gpurun_out/r04_w2_segments.md > gpurun_out/r04_w2_segments.log 2>&1
for wl in usss_g wsss; do python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_bench_$wl.json 2>/dev/null; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-prof --force-exchange > gpurun_out/r04_bench_forced_nccl.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-prof --graph > gpurun_out/r04_bench_graph.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-alt --no-prof > gpurun_out/r04_bench_2rank_gloo.json 2>/dev/null
ls gpurun_out | grep r04 | head -50
