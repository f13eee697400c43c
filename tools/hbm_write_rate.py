"""What the part sustains for write-dominated streams: fill (write only), copy (read + write) and read-only reductions over
buffers smaller and larger than the 256 MiB Infinity Cache, timed with HIP events (torch kernels, no library code involved).
The stft / power_spectrum outputs of the path are write-dominated: this is the ceiling they run against."""
import torch


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n


for mb in (64, 128, 192, 269, 384, 512, 1024, 2048):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    t_fill = timed(lambda: a.fill_(1.0))
    t_copy = timed(lambda: b.copy_(a))
    t_read = timed(lambda: a.sum())
    # two buffers written alternately (what bench.py's output ring does): footprint 2 x mb
    t_fill2 = timed(lambda: (a.fill_(1.0), b.fill_(2.0))) / 2
    print("%5d MB: fill %.2f TB/s | fill, two buffers alternating %.2f TB/s | copy %.2f TB/s (read + write) | sum %.2f TB/s"
          % (mb, mb * 1e6 / t_fill / 1e12, mb * 1e6 / t_fill2 / 1e12, 2 * mb * 1e6 / t_copy / 1e12, mb * 1e6 / t_read / 1e12))
