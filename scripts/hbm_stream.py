"""Measured HBM rates on the box (SURVEY 8d asks for measured peaks beside the datasheet's): device-to-device
copy, fill (write-only) and triad a = b + s*c over 4 GiB arrays, torch elementwise kernels, HIP events."""
import torch
dev = torch.device('cuda', 0)
n = 1 << 29                       # 4 GiB of float64 per array
a = torch.empty(n, dtype=torch.float64, device=dev)
b = torch.ones(n, dtype=torch.float64, device=dev)
c = torch.ones(n, dtype=torch.float64, device=dev)
def timed(fn, bytes_moved, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return bytes_moved / ms / 1e6     # GB/s
print('copy  (read+write): %.0f GB/s' % timed(lambda: a.copy_(b), 16 * n))
print('fill  (write only): %.0f GB/s' % timed(lambda: a.fill_(1.5), 8 * n))
print('triad (2 read + 1 write): %.0f GB/s' % timed(lambda: torch.add(b, c, alpha=2.0, out=a), 24 * n))
print('sum   (read only): %.0f GB/s' % timed(lambda: b.sum(), 8 * n))
