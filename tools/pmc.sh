# usage: pmc.sh <outname> "<counters>" <bench_conv filter...>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
CTRS="$1"; shift
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o pmc -- python tools/bench_conv.py "$@" > $OUT/run.log 2>&1
tail -4 $OUT/run.log | cut -c1-200
ls $OUT | head
