"""Infinity-Cache probe: effective bandwidth of a streaming copy and of a producer -> consumer pair as a function of the buffer size."""
import torch, sys
dev = torch.device('cuda:0')
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3
big = torch.empty(3 * 1024**3 // 4, dtype=torch.float32, device=dev)   # 3 GiB scratch to flush the cache between pairs
for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
    n = mb * 1024 * 1024 // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a); c = torch.empty_like(a)
    t_copy = timeit(lambda: b.copy_(a))
    # producer -> consumer: b = a * 2 (writes b), then c = b + 1 (reads b just written); time only the consumer by differencing
    def pair():
        torch.mul(a, 2.0, out=b); torch.add(b, 1.0, out=c)
    def prod():
        torch.mul(a, 2.0, out=b)
    t_pair, t_prod = timeit(pair), timeit(prod)
    print('%5d MB  copy %6.2f TB/s   producer %6.2f TB/s   consumer-after-producer %6.2f TB/s' % (
        mb, 2 * n * 4 / t_copy / 1e12, 2 * n * 4 / t_prod / 1e12, 2 * n * 4 / max(t_pair - t_prod, 1e-9) / 1e12))
    del a, b, c
