# HBM traffic of the dominant kernel family over the real bench step: two PMC passes
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass), kernel-trace only.
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_bench_$pass
  rm -rf $OUT && mkdir -p $OUT
  ( cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o pmc -- \
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-alt > $OUT/run.log 2>&1 )
  tail -1 $OUT/run.log | cut -c1-200
  # the raw CSVs are large: keep only the per-kernel aggregate
  python3 - "$OUT" "$pass" <<'PY'
import csv, sys, collections, json
d, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(d + '/pmc_counter_collection.csv')):
    if r['Counter_Name'] != ctr:
        continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:90]
    agg[k][0] += 1
    agg[k][1] += float(r['Counter_Value'])
json.dump({k: {'dispatches': v[0], ctr + '_KiB_total': v[1]} for k, v in agg.items()}, open(d + '/agg.json', 'w'), indent=1)
print(len(agg), 'kernels aggregated')
PY
  rm -f $OUT/pmc_counter_collection.csv $OUT/pmc_kernel_trace.csv
done
