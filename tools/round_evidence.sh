# Round evidence on the GPU box (one gpurun call): bench + per-layer tables + kernel trace, PMC (split GEMM, fused F(2x2), HBM traffic with
# calibration), the Generator-step decision probe, the other workloads and launch modes.  usage: R=r05 bash tools/round_evidence.sh
R=${R:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/evidence.sh $R notests 2>&1 | tail -15
bash tools/pmc_split_gemm.sh $R > gpurun_out/${R}_pmc_sg.log 2>&1; tail -3 gpurun_out/${R}_pmc_sg.log
bash tools/pmc_wino2.sh $R > gpurun_out/${R}_pmc_w2.log 2>&1; tail -3 gpurun_out/${R}_pmc_w2.log
ROUND=$R bash tools/pmc_hbm.sh > gpurun_out/${R}_pmc_hbm.log 2>&1; tail -3 gpurun_out/${R}_pmc_hbm.log
python tools/parity_probe_g.py > gpurun_out/${R}_parity_probe_g.log 2>&1; cp gpurun_out/parity_probe_g.json gpurun_out/${R}_parity_probe_g.json
for wl in usss_g wsss; do python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${R}_bench_$wl.json 2>/dev/null; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-prof --force-exchange > gpurun_out/${R}_bench_forced_nccl.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --batch 4 --no-cpu-baseline --no-alt --no-prof > gpurun_out/${R}_bench_2rank_gloo.json 2>/dev/null
FCD_WINO_CHAIN=0 FCD_WINO_XCD2=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --layers-md gpurun_out/${R}_layers_r4paths.md > gpurun_out/${R}_bench_r4paths.json 2>/dev/null
python bench.py --steps 20 --warmup 3 > gpurun_out/${R}_bench_final.json 2>/dev/null; tail -c 400 gpurun_out/${R}_bench_final.json
ls gpurun_out | grep $R | head -50
