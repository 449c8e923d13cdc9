# rocprofv3 kernel trace of tools/bn_probe.py (BatchNorm kernels on the Segmentor's layer shapes); FCD_LIB selects an A/B build
cd /tmp && export TMPDIR=/tmp
TAG=${1:-new}
OUT=$GRAFT_REPO_ROOT/gpurun_out/bnprof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $OUT -o bn -- python tools/bn_probe.py > $OUT/probe.log 2>&1
DB=$(find $OUT -name '*.db' | head -1); python tools/rocpd_stats.py $DB $OUT/trace.md > /dev/null 2>&1; echo $TAG; grep -E "bn_" $OUT/trace.md | cut -c1-60,130-200 | head -5
rm -f $DB
