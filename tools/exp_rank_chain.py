#!/usr/bin/env python3
"""Ordered joins of a small build side under a long probe side (the sort on (rank, payload) composites): its LSD passes as exact
passes (hist + scan + scatter, 48 B per row and pass: HMJ_GTABLE_SORT_SLAB=0) against the chain of histogram-free slab passes
(32 B).  The result columns of the two are compared element by element.  usage: exp_rank_chain.py [log2 probe rows ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hashmergejoin_amd as H

os.environ["HMJ_GTABLE_SORT_FANOUT"] = "1"
os.environ["HMJ_GTABLE_SORT_SLAB"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_GTABLE_SORT_SLAB"] = "1"
ex1 = H.Executor(0)


def timed(e, R, S, fl, reps=3):
    for _ in range(2):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


for P in [int(x) for x in sys.argv[1:]] or [26]:
    for k in (8, 10, 12, 13, 14, 16, 18, 20):
        if k + 4 > P:
            continue
        R, S = ex0.gen_build((1 << k) - (k % 3) * 37), ex0.gen_uniform_domain(1 << P, (1 << k) - (k % 3) * 37)
        fl = H.HMJ_ORDERED | H.HMJ_CHECKSUM
        m0, r0 = timed(ex0, R, S, fl)
        p0 = ex0.last_timing()
        cols0 = ex0.columns_to_numpy(r0, host=False) if (P <= 26 and k in (10, 14, 18)) else None
        ck0 = r0.checks()
        ex0.release_result()
        m1, r1 = timed(ex1, R, S, fl)
        p1 = ex1.last_timing()
        ok = ck0 == r1.checks()
        if cols0 is not None:
            ok = ok and bool((cols0 == ex1.columns_to_numpy(r1, host=False)).all())
        print("nb=%d np=2^%d | exact passes %.3f ms (%d bits, %d passes, path %#x) | slab chain %.3f ms (%d passes, path %#x)%s" % (
            (1 << k) - (k % 3) * 37, P, m0, p0["radix_bits"], p0["radix_passes"], p0["path"], m1, p1["radix_passes"], p1["path"], "" if ok else " MISMATCH"), flush=True)
        ex1.release_result()
        del R, S, cols0
