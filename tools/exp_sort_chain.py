#!/usr/bin/env python3
"""hmj_sort_u64_device: exact passes (HMJ_SORT_SLAB=0) against the chain of slab passes + compaction.
usage: exp_sort_chain.py [log2 rows ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

os.environ["HMJ_SORT_SLAB"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_SORT_SLAB"] = "1"
ex1 = H.Executor(0)


def timed(e, a, reps=3):
    for _ in range(2):
        out = e.sort_device(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = e.sort_device(a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


for lg in [int(x) for x in sys.argv[1:]] or [26, 28]:
    n = 1 << lg
    for name in ("uniform 64-bit keys", "dense ids (a permutation of 0 .. n-1)"):
        if name.startswith("uniform"):
            a = ex0.gen_build(n)  # mix64(i): uniform over 64 bits
        else:
            a = torch.stack([torch.randperm(n, device="cuda"), torch.arange(n, device="cuda")], 1).contiguous()
        m0, o0 = timed(ex0, a)
        m1, o1 = timed(ex1, a)
        ok = torch.equal(o0, o1)
        print("n=2^%d %s | exact passes %.3f ms = %.1f G keys/s | slab chain %.3f ms = %.1f G keys/s (path %#x)%s" % (
            lg, name, m0, n / m0 * 1e-6, m1, n / m1 * 1e-6, ex1.last_timing()["path"], "" if ok else " MISMATCH"), flush=True)
        del a, o0, o1
