# Round evidence on the GPU box (gpurun): GPU test suite, bench + per-layer tables, rocprofv3 kernel trace of the bench command.
# usage: bash tools/evidence.sh r03 [notests]
R=${1:-r03}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/${R}_pytest_gpu.txt 2>&1; tail -3 gpurun_out/${R}_pytest_gpu.txt
fi
python bench.py --steps 10 --warmup 3 --layers-md gpurun_out/${R}_layers.md > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err; tail -c 300 gpurun_out/${R}_bench.json
bash tools/prof.sh ${R}_prof > gpurun_out/${R}_prof.log 2>&1; DB=$(find gpurun_out/${R}_prof -name '*.db' | head -1); python tools/rocpd_stats.py $DB gpurun_out/${R}_kernel_trace.md > /dev/null 2>&1; head -12 gpurun_out/${R}_kernel_trace.md
rm -rf gpurun_out/${R}_prof
