"""Development probe: what a plain linear fill / copy reaches on this chip (the practical ceiling next to the 8 TB/s of the data sheet).
    python scripts/attic/hbm_ceiling_probe.py"""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for gb in (0.4, 1.6, 3.2):
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    t = timeit(lambda: x.zero_())
    print("zero_  %.1f GB: %.3f ms  %.2f TB/s" % (gb, t, gb / t))
    t = timeit(lambda: x.fill_(1.5))
    print("fill_  %.1f GB: %.3f ms  %.2f TB/s" % (gb, t, gb / t))
    t = timeit(lambda: y.copy_(x))
    print("copy_  %.1f GB: %.3f ms  %.2f TB/s (read+write %.2f)" % (gb, t, gb / t, 2 * gb / t))
    del x, y
