#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache absorb a recycled scratch buffer?  Ring copies a->b->c->a at several
footprints: if small rings run much faster per byte than a 4 GiB copy, writes stay on-die."""
import torch, time
dev = torch.device("cuda:0")
def run(mb, iters):
    n = mb * (1 << 20) // 8
    bufs = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(3)]
    bufs[0].random_()
    for i in range(3): bufs[(i + 1) % 3].copy_(bufs[i % 3])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): bufs[(i + 1) % 3].copy_(bufs[i % 3])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("ring of 3 x %5d MiB: %.4f ms per copy, %.2f TB/s (read+write)" % (mb, ms, 2 * mb * (1 << 20) / ms / 1e9), flush=True)
for mb in [8, 16, 32, 48, 64, 96, 128, 256, 1024, 4096]:
    run(mb, max(4, min(200, 8192 // mb)))
