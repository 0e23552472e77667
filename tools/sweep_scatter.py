#!/usr/bin/env python3
"""GPU microbench: one radix pass (hist/scan/scatter) on 2^log2n rows vs fan-out bits and shift."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ex = H.Executor(0)
ex.set_profiling(True)
n = 1 << log2n
R = ex.gen_build(n)
for bits, shift in [(1, 63), (3, 61), (5, 59), (6, 58), (7, 57), (8, 56), (9, 55), (8, 47), (9, 47), (8, 20)]:
    best = None
    for it in range(3):
        out, off = ex.partition_device(R, shift, bits)
        t = ex.last_timing()
        if best is None or t["ms_scatter"] < best["ms_scatter"]:
            best = t
        del out, off
    gbs = 32.0 * n / (best["ms_scatter"] * 1e-3) / 1e9
    print("bits=%d shift=%2d  hist %.3f ms  scatter %.3f ms  (%.0f GB/s alg)" % (bits, shift, best["ms_hist"], best["ms_scatter"], gbs), flush=True)
