#!/usr/bin/env python3
"""Ordered joins of a small build side under a long probe side: the sort on (rank, payload) composites against the partitioned
paths (HMJ_GTABLE_SORT=0).  usage: exp_small_ordered.py [log2 probe rows = 26]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 26
os.environ["HMJ_GTABLE_SORT"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_GTABLE_SORT"] = "1"
os.environ["HMJ_GTABLE_SORT_FANOUT"] = "1"
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


for k in range(8, min(P - 3, 25), 1):
    R, S = ex0.gen_build(1 << k), ex0.gen_uniform_domain(1 << P, 1 << k)
    fl = H.HMJ_ORDERED | H.HMJ_CHECKSUM
    m0, r0 = timed(ex0, R, S, fl)
    p0 = ex0.last_timing()
    m1, r1 = timed(ex1, R, S, fl)
    p1 = ex1.last_timing()
    ok = r0.checks() == r1.checks()
    print("nb=2^%d np=2^%d fan-out %d | partitioned %.3f ms (b%d path %#x) | rank sort %.3f ms (%d bits, path %#x)%s" % (
        k, P, 1 << (P - k), m0, p0["radix_bits"], p0["path"], m1, p1["radix_bits"], p1["path"], "" if ok else " MISMATCH"), flush=True)
    ex0.release_result()
    ex1.release_result()
    del R, S
