#!/usr/bin/env python3
"""Derived figures from the per-kernel counter means of tools/pmc_split_gemm.sh / tools/pmc_wino2.sh (two SQ passes each).

    python tools/pmc_md.py gpurun_out/r04_pmc_split_gemm.txt 32 > profiles/r04_pmc_split_gemm.md     # 32 cycles per bf16 32x32x16 MFMA
    python tools/pmc_md.py gpurun_out/r04_pmc_wino2.txt 32 > profiles/r04_pmc_wino2.md               # 32 cycles per fp32 16x16x4 MFMA

Per kernel (and launch grid): launch time (kernel trace of the same run), shader clock = GRBM_GUI_ACTIVE / 8 XCDs / time, MFMA-busy
(SQ_VALU_MFMA_BUSY_CYCLES over 1024 SIMDs x active cycles), the clock-adjusted view of the roofline fraction the bench line prices
at the nominal 2.4 GHz, instruction mix per MFMA, wait shares, LDS bank conflicts."""
import collections
import re
import sys

path = sys.argv[1]
blocks = collections.OrderedDict()
cur = None
for line in open(path):
    m = re.match(r'^(\S.*?)\s+dispatches\s+(\d+)\s*$', line)
    if m:
        cur = blocks.setdefault(m.group(1).strip(), {})
        cur['dispatches'] = int(m.group(2))
        continue
    m = re.match(r'^\s+(\w+)\s+([0-9.eE+]+)\s*$', line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))

NOMINAL_GHZ = 2.4
print('# PMC counters, derived (tools/pmc_md.py over %s)' % path.split('/')[-1])
print()
print('Two rocprofv3 passes of 8 SQ counters each (`--pmc ... --kernel-trace`, no other trace domain); per-launch means.  Clock = GRBM_GUI_ACTIVE / 8 XCDs /')
print('launch time of the same run; MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles) = the fraction of the matrix-pipe peak AT THE CLOCK THE')
print('KERNEL RAN AT (the clock-adjusted roofline fraction); times clock / %.1f GHz it is the fraction of the nominal peak the bench line reports.  The gap between' % NOMINAL_GHZ)
print('the two columns is power management (the chip lowers its clock under dense bf16 MFMA work), the gap between MFMA-busy and 100 % is the instruction stream.')
print()
print('| kernel | launch us | active cycles / XCD | clock GHz | MFMA-busy (= of peak at the measured clock) | of the nominal %.1f-GHz peak | VALU per MFMA | LDS instr per MFMA | s_waitcnt / barrier share of wave cycles | waiting-to-issue share | LDS bank-conflict cycles / LDS active |' % NOMINAL_GHZ)
print('|---|---|---|---|---|---|---|---|---|---|---|')
for name, c in blocks.items():
    if 'GRBM_GUI_ACTIVE' not in c:
        continue
    cyc = c['GRBM_GUI_ACTIVE'] / 8.0
    dur_us = c.get('DURATION_NS', 0.0) / 1e3
    ghz = cyc / (dur_us * 1e3) if dur_us else float('nan')
    busy = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc)
    mf = c.get('SQ_INSTS_MFMA', 0.0)
    valu = (c.get('SQ_INSTS_VALU', 0.0) - mf) / mf if mf else float('nan')
    lds = c.get('SQ_INSTS_LDS', 0.0) / mf if mf else float('nan')
    wa = c.get('SQ_WAIT_ANY', 0.0) / c['SQ_WAVE_CYCLES']
    wi = c.get('SQ_WAIT_INST_ANY', 0.0) / c['SQ_WAVE_CYCLES']
    bc = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE'] if c.get('SQ_LDS_IDX_ACTIVE') else float('nan')
    print('| %s | %.1f | %.3g | %.2f | %.1f %% | %.1f %% | %.2f | %.2f | %.0f %% | %.0f %% | %.1f %% |' % (
        name, dur_us, cyc, ghz, 100 * busy, 100 * busy * ghz / NOMINAL_GHZ, valu, lds, 100 * wa, 100 * wi, 100 * bc))
