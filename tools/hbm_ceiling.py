"""measured HBM ceiling of the box: device-to-device copy and a read-only reduction (torch kernels), GB/s"""
import time, torch
torch.cuda.set_device(0)
n = 1 << 30                                   # 4 GiB of f32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
for name, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 2 * a.numel() * 4), ("sum (read)", lambda: a.sum(), a.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%s: %.0f GB/s" % (name, bytes_moved / dt / 1e9))
