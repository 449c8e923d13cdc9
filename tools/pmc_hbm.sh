#!/bin/bash
# HBM traffic of the step's kernels from the PMC counters, corrected with a calibration measured in the same session:
#   1. tools/hbm_calib.bin (known 2 GiB streams in our access patterns) under --pmc FETCH_SIZE and --pmc WRITE_SIZE
#   2. bench.py (2 steps + 1 warm-up) under the same two passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: separate
#      passes; --kernel-trace only, as MI355X_MICROARCH.md prescribes)
#   3. tools/pmc_hbm_report.py -> gpurun_out/${ROUND}_hbm_traffic.json (stamped with the kernel source hash; copy to profiles/)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r03}
agg() {  # dir counter
python3 - "$1" "$2" <<'PY'
import csv, sys, collections, json, glob
d, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != ctr:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][:100]
        agg[k][0] += 1
        agg[k][1] += float(r['Counter_Value'])
json.dump({k: {'dispatches': v[0], ctr: v[1]} for k, v in agg.items()}, open(d + '/agg.json', 'w'), indent=1)
print(ctr, len(agg), 'kernels aggregated')
PY
find "$1" -name '*.csv' -delete
}
[ -x $ROOT/tools/hbm_calib.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $ROOT/tools/hbm_calib.hip -o $ROOT/tools/hbm_calib.bin   # (the binary is git-ignored)
for pass in FETCH_SIZE WRITE_SIZE; do
  OUT=$ROOT/gpurun_out/pmc_calib_$pass; rm -rf $OUT; mkdir -p $OUT
  timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o c -- $ROOT/tools/hbm_calib.bin > $OUT/run.log 2>&1
  tail -1 $OUT/run.log | cut -c1-200
  agg $OUT $pass
  OUT=$ROOT/gpurun_out/pmc_bench_$pass; rm -rf $OUT; mkdir -p $OUT
  ( cd $ROOT && timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o pmc -- \
      python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-prof --no-alt > $OUT/run.log 2>&1 )
  tail -1 $OUT/run.log | cut -c1-200
  agg $OUT $pass
done
cd $ROOT && python3 tools/pmc_hbm_report.py gpurun_out > gpurun_out/${ROUND}_hbm_traffic.json && head -c 1500 gpurun_out/${ROUND}_hbm_traffic.json
