#!/usr/bin/env python3
"""Which ATen operators run on the hot path of one Demo_RSSS step (bench.py's headline workload)?  torch.profiler over ONE step,
CPU-side operator names with their call counts and the device kernels they launch -- the glue the fcd kernels do not cover
(VERDICT r4 item 7)."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench            # noqa: E402


def main():
    import argparse
    _, bands, size, batch, _ = bench.WORKLOADS['rsss']
    args = argparse.Namespace(workload='rsss', bands=bands, size=size, batch=batch)
    torch.cuda.set_device(0)
    step, _ = bench.build_workload(args, torch.device('cuda', 0), 0)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith('aten::') and e.device_time_total > 0]
    rows.sort(key=lambda e: -e.device_time_total)
    for e in rows[:60]:
        stack = [s for s in e.stack if 'fcd_gan_pytorch_amd' in s or 'bench.py' in s][:2]
        print('%-28s calls %4d  dev %8.1f us  %s' % (e.key, e.count, e.device_time_total, ' <- '.join(s.split('/')[-1][:60] for s in stack)))


if __name__ == '__main__':
    main()
