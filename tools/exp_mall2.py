#!/usr/bin/env python3
"""Are Infinity-Cache traffic and HBM traffic additive?  big->small (HBM reads, on-die writes),
small->big (on-die reads, HBM writes), big->big, and a read-only sweep."""
import torch
dev = torch.device("cuda:0")
GB = 1 << 30
big = torch.empty(4 * GB // 8, dtype=torch.int64, device=dev); big.random_()
big2 = torch.empty(4 * GB // 8, dtype=torch.int64, device=dev)
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
for mb in [32, 64]:
    n = mb * (1 << 20) // 8
    chunks = big.view(-1, n); chunks2 = big2.view(-1, n)
    small = torch.empty(n, dtype=torch.int64, device=dev)
    k = chunks.shape[0]
    def b2s():
        for i in range(k): small.copy_(chunks[i])
    def s2b():
        for i in range(k): chunks2[i].copy_(small)
    def b2b():
        for i in range(k): chunks2[i].copy_(chunks[i])
    def b2s2b():  # big -> small -> big2: the fused-pass shape (HBM read 4G, HBM write 4G, scratch on die)
        for i in range(k):
            small.copy_(chunks[i]); chunks2[i].copy_(small)
    for name, fn, hbm in [("big->small", b2s, 4), ("small->big", s2b, 4), ("big->big", b2b, 8), ("big->small->big", b2s2b, 8)]:
        ms = timed(fn)
        print("chunk %3d MiB %-16s %.3f ms  -> %.2f TB/s of HBM-side bytes" % (mb, name, ms, hbm * GB / ms / 1e9), flush=True)
ms = timed(lambda: big.sum())
print("read-only sum of 4 GiB: %.3f ms -> %.2f TB/s" % (ms, 4 * GB / ms / 1e9))
ms = timed(lambda: big2.fill_(7))
print("write-only fill of 4 GiB: %.3f ms -> %.2f TB/s" % (ms, 4 * GB / ms / 1e9))
