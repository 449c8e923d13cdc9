#!/bin/bash
# SQ counters of the F(4x4) batched GEMM on VGG conv4_x / conv3_x / conv2_2: split-bf16 kernels (default) and the fp32
# kernel (FCD_WINO_SPLIT=0) beside them; two passes of 8 SQ counters each; summary -> gpurun_out/<tag>_pmc_split_gemm.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02}
rm -f $ROOT/gpurun_out/${TAG}_pmc_split_gemm.txt
for split in 1 0; do
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  OUT=$ROOT/gpurun_out/pmc_sg_${split}_$(echo $pass | cut -d' ' -f1); rm -rf $OUT; mkdir -p $OUT
  ( cd $ROOT && FCD_WINO_SPLIT=$split timeout 300 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT -o pmc -- \
      python tools/bench_wino_gemm.py "vgg 512->512 @32" "vgg 256->256 @64" "vgg 128->128 @128" > $OUT/run.log 2>&1 )
  tail -4 $OUT/run.log | cut -c1-110
  python3 - "$OUT" "$split" <<'PY' >> $ROOT/gpurun_out/${TAG}_pmc_split_gemm.txt
import csv, sys, collections, glob, re
d, split = sys.argv[1], sys.argv[2]
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(wino_gemm\w*kernel(?:<[^>]*>)?)', r['Kernel_Name'])
        if m:
            # one row per (kernel, grid): the three layers launch different grids
            k = '%s grid=%s' % (m.group(1), r.get('Grid_Size', r.get('Grid_Size_X', '?')))
            ctr[k][r['Counter_Name']].append(float(r['Counter_Value']))
# launch durations of the same run (kernel trace): with GRBM_GUI_ACTIVE they give the shader clock the kernel really ran at
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(wino_gemm\w*kernel(?:<[^>]*>)?)', r['Kernel_Name'])
        if m:
            # the trace names the grid per dimension (Grid_Size_X/Y/Z or Grid_Size), the counter file as one number
            if 'Grid_Size' in r:
                gs = r['Grid_Size']
            else:
                gs = str(int(r.get('Grid_Size_X', 1)) * int(r.get('Grid_Size_Y', 1)) * int(r.get('Grid_Size_Z', 1)))
            k = '%s grid=%s' % (m.group(1), gs)
            if k not in ctr:          # fall back to the kernel name when there is exactly one grid of it
                cand = [q for q in ctr if q.startswith(m.group(1) + ' grid=')]
                k = cand[0] if len(cand) == 1 else None
            if k:
                ctr[k]['DURATION_NS'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
for k, cs in sorted(ctr.items()):
    print('FCD_WINO_SPLIT=%s  %s  dispatches %d' % (split, k, max(len(v) for v in cs.values())))
    for c, v in sorted(cs.items()):
        print('    %-28s %18.0f' % (c, sum(v) / len(v)))
PY
  find $OUT -name '*.csv' -delete
done
done
cat $ROOT/gpurun_out/${TAG}_pmc_split_gemm.txt
