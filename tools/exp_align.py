#!/usr/bin/env python3
"""Experiment: which run ALIGNMENT does the scatter need?  Keys crafted so every tile holds a fixed
number of rows of each digit -> every run is line-aligned (128 B), or only 64-B / 32-B aligned."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
ex = H.Executor(0); ex.set_profiling(True)
n = 1 << 28
i = torch.arange(n, dtype=torch.int64, device="cuda")
bits = 9
def run(name, dig):
    R = torch.stack([dig << (64 - bits), i], 1).contiguous()
    best = 1e9
    for _ in range(3):
        out, off = ex.partition_device(R, 64 - bits, bits)
        best = min(best, ex.last_timing()["ms_scatter"]); del out, off
    print("bits=%d %-28s scatter %.3f ms (%.0f GB/s)" % (bits, name, best, 32.0 * n / best / 1e6), flush=True)
g = (i // 16) % 256
r = i % 16
run("128B-aligned runs of 8", (i // 8) % 512)
run("64B-aligned runs 4/12", torch.where(r < 4, 2 * g, 2 * g + 1))
run("32B-aligned runs 2/14", torch.where(r < 2, 2 * g, 2 * g + 1))
run("16B-aligned runs 3/13", torch.where(r < 3, 2 * g, 2 * g + 1))
run("16B-aligned runs 7/9", torch.where(r < 7, 2 * g, 2 * g + 1))
