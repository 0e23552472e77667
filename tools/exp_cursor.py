#!/usr/bin/env python3
"""Unordered materialising joins that place rows behind ONE result cursor (global table: gtable_write_kernel; mid-size build
sides: probe_kernel<3> over the slabs of one pass) against the partitioned count / scan / write passes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

os.environ["HMJ_GTABLE"] = "0"
os.environ["HMJ_ONE_PASS_SLAB"] = "0"
ex0 = H.Executor(0)
del os.environ["HMJ_GTABLE"], os.environ["HMJ_ONE_PASS_SLAB"]
ex1 = H.Executor(0)


def timed(e, R, S, fl, reps=5):
    for _ in range(2):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


for P in (26, 28):
    for k in (10, 14, 16, 17, 18, 19, 20, 21):
        R, S = ex0.gen_build(1 << k), ex0.gen_uniform_domain(1 << P, 1 << k)
        row = []
        for fl, name in ((H.HMJ_MATERIALIZE | H.HMJ_CHECKSUM, "rows"), (H.HMJ_MATERIALIZE | H.HMJ_FIRST_WINS | H.HMJ_CHECKSUM, "rows, first-wins")):
            m0, r0 = timed(ex0, R, S, fl)
            c0 = r0.checks()
            m1, r1 = timed(ex1, R, S, fl)
            ok = c0 == r1.checks()
            row.append("%s: partitioned %.3f ms (b%d) | cursor %.3f ms (path %#x)%s" % (name, m0, ex0.last_timing()["radix_bits"], m1, ex1.last_timing()["path"], "" if ok else " MISMATCH"))
        print("build 2^%d x probe 2^%d | %s" % (k, P, " | ".join(row)), flush=True)
        ex0.release_result()
        ex1.release_result()
        del R, S
