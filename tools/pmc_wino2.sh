#!/bin/bash
# SQ counters of the fused F(2x2) kernel on VGG conv1_2 (two passes: 8 SQ slots each); summary -> gpurun_out/<tag>_pmc_wino2.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  OUT=$ROOT/gpurun_out/pmc_w2_$(echo $pass | cut -d' ' -f1); rm -rf $OUT; mkdir -p $OUT
  ( cd $ROOT && W2_ONLY=1 timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o pmc -- python tools/bench_wino2.py > $OUT/run.log 2>&1 )
  tail -2 $OUT/run.log | cut -c1-200
  python3 - "$OUT" <<'PY' >> $ROOT/gpurun_out/${TAG}_pmc_wino2.txt
import csv, sys, collections, glob
d = sys.argv[1]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m = re.search(r'(conv_wino2_kernel<[^>]*>|conv_igemm_glds_kernel<[^>]*>)', r['Kernel_Name'])
        k = m.group(1) if m else ''
        if k:
            ctr[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(conv_wino2_kernel<[^>]*>|conv_igemm_glds_kernel<[^>]*>)', r['Kernel_Name'])
        if m and m.group(1) in ctr:
            ctr[m.group(1)]['DURATION_NS'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, cs in ctr.items():
    print(k, ' dispatches', max(len(v) for v in cs.values()))
    for c, v in sorted(cs.items()):
        print('    %-28s %18.0f' % (c, sum(v) / len(v)))
PY
  find $OUT -name '*.csv' -delete
done
cat $ROOT/gpurun_out/${TAG}_pmc_wino2.txt
