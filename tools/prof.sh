# rocprofv3 kernel-trace + stats of the default bench command (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prof > $OUT/bench.log 2>&1
tail -2 $OUT/bench.log | cut -c1-400
find $OUT -name "*stats*" | head
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows),'kernels')
for r in rows[:28]:
    print('%-150s calls %6s total_ms %9.2f avg_us %9.1f pct %5s' % (r['Name'][:150], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, r['Percentage']))
PY
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +20M -delete
