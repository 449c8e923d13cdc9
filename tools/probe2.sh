cd $GRAFT_REPO_ROOT
python - <<'PY'
import os
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max','/sys/fs/cgroup/cpu/cpu.cfs_quota_us','/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
PY
grep -m1 "model name" /proc/cpuinfo
for t in 8 16 32 64; do
timeout 120 python - <<PY
import torch, time, torch.nn.functional as F
torch.set_num_threads($t)
x=torch.randn(2,64,256,256); w=torch.randn(64,64,3,3)
F.conv2d(x,w,padding=1)
t0=time.time()
for _ in range(3): F.conv2d(x,w,padding=1)
dt=(time.time()-t0)/3
print('threads',$t,'conv 64->64 256^2 N=2: %.3fs  %.2f TFLOP/s'%(dt, 2*2*64*64*9*65536/dt/1e12), flush=True)
PY
done
