#!/usr/bin/env python3
"""One join shape, timed, with the plan and the path it took.  usage: exp_one.py log2_build log2_probe [flags: ordered|count|mat] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

kb, kp = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "ordered"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
fl = {"ordered": H.HMJ_ORDERED | H.HMJ_CHECKSUM, "count": H.HMJ_CHECKSUM, "mat": H.HMJ_MATERIALIZE | H.HMJ_CHECKSUM}[mode]
ex = H.Executor(0)
R, S = ex.gen_build(1 << kb), ex.gen_uniform_domain(1 << kp, 1 << kb)
for i in range(reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ex.join_device(R, S, fl)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    t = ex.last_timing()
    print("2^%d x 2^%d %s: %.3f ms path %#x bits %d passes %d plan %s" % (kb, kp, mode, ms, t["path"], t["radix_bits"], t["radix_passes"], ex.last_plan()), flush=True)
