#!/usr/bin/env python3
"""Count joins of a mid-size build side (2^17 ... 2^21 rows) under a big probe side: the one-pass slab plan (pass-A slabs probed
piece by piece) against the exact one-pass plan (HMJ_ONE_PASS_SLAB=0).  usage: exp_midbuild.py [log2 probe rows = 26]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 26
os.environ["HMJ_ONE_PASS_SLAB"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_ONE_PASS_SLAB"] = "1"
ex1 = H.Executor(0)


def timed(e, R, S, fl, reps=6):
    for _ in range(2):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    e.set_profiling(True)
    e.join_device(R, S, fl)
    t = e.last_timing()
    e.set_profiling(False)
    return ms, r, t


n_p = 1 << P
for k in (17, 18, 19, 20, 21):
    for mul in (1.0, 1.4):
        nb = int((1 << k) * mul)
        R = ex0.gen_build(nb)
        S = ex0.gen_uniform_domain(n_p, nb)
        for fl, name in ((0, "count"), (H.HMJ_CHECKSUM | H.HMJ_SUM_PROBE, "checks")):
            m0, r0, t0 = timed(ex0, R, S, fl)
            m1, r1, t1 = timed(ex1, R, S, fl)
            ok = r0.checks() == r1.checks() and int(r0.sum_probe_all) == int(r1.sum_probe_all)
            print("nb=%8d np=2^%d %-6s | exact %.3f ms (b%d path %#x part %.3f probe %.3f) | one-pass slab %.3f ms (path %#x part %.3f probe %.3f)%s" % (
                nb, P, name, m0, t0["radix_bits"], t0["path"], t0["ms_partition_build"] + t0["ms_partition_probe"], t0["ms_probe_count"],
                m1, t1["path"], t1["ms_partition_build"] + t1["ms_partition_probe"], t1["ms_probe_count"], "" if ok else "  MISMATCH"), flush=True)
        del R, S
