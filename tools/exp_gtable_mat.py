#!/usr/bin/env python3
"""Materialising (unordered) joins of a small build side: global table + ballot-compacted writes against the partitioned paths."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 26
os.environ["HMJ_GTABLE"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_GTABLE"] = "1"
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


for k in range(10, 18):
    R, S = ex0.gen_build(1 << k), ex0.gen_uniform_domain(1 << P, 1 << k)
    row = []
    for fl, name in ((H.HMJ_MATERIALIZE | H.HMJ_CHECKSUM, "mat"), (H.HMJ_MATERIALIZE | H.HMJ_FIRST_WINS | H.HMJ_CHECKSUM, "mat+first")):
        m0, r0 = timed(ex0, R, S, fl)
        m1, r1 = timed(ex1, R, S, fl)
        ok = r0.checks() == r1.checks() and bool(ex1.last_timing()["path"] & H.HMJ_PATH_GLOBAL_TABLE)
        row.append("%s partitioned %.3f ms (b%d) | global table %.3f ms%s" % (name, m0, ex0.last_timing()["radix_bits"], m1, "" if ok else " MISMATCH"))
    print("nb=2^%d np=2^%d | %s" % (k, P, " | ".join(row)), flush=True)
    ex0.release_result()
    ex1.release_result()
