#!/usr/bin/env python3
"""hmj_sort_u64_device out of place over sizes (uniform 64-bit keys): ms, G keys/s, path.  usage: exp_sort_sizes.py [n ...] (rows; default 2^26 ... 10^9)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

sizes = [int(float(x)) for x in sys.argv[1:]] or [1 << 26, 1 << 28, 500_000_000, 1 << 29, 1_000_000_000, 1 << 30]
ex = H.Executor(0)
for n in sizes:
    R = ex.gen_build(n)
    best = None
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = ex.sort_device(R)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if i and (best is None or dt < best):
            best = dt
    t = ex.last_timing()
    ok = None
    if n <= (1 << 29):  # ascending as UNSIGNED keys
        k = out[:, 0] ^ torch.iinfo(torch.int64).min
        ok = bool((k[1:] >= k[:-1]).all())
        del k
    print("n=%d (2^%.2f) uniform  %.3f ms  %.1f G keys/s  path %#x bits %d sorted=%s" % (n, __import__("math").log2(n), best, n / best * 1e-6, t["path"], t["radix_bits"], ok), flush=True)
    del R, out
    ex.release_result()
