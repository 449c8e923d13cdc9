# rocprofv3 kernel-trace + stats of the default bench command (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace -d $OUT -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof --no-alt > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-300; ls -la $OUT
