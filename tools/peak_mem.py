"""Peak device memory of the headline step (framework allocator statistics)."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ap = argparse.Namespace(workload='rsss', bands=13, size=256, batch=int(os.environ.get('BATCH', '8')))
step, _ = bench.build_workload(ap, torch.device('cuda', 0), 0)
for _ in range(3):
    step()
torch.cuda.synchronize()
print('batch', ap.batch, 'peak allocated %.1f GB, reserved %.1f GB' % (torch.cuda.max_memory_allocated() / 2**30, torch.cuda.max_memory_reserved() / 2**30))
