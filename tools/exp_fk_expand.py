#!/usr/bin/env python3
"""Ordered foreign-key joins (unique build keys, f probe rows per key): the one-pass ordered write (in-run ranking linear in f)
against count + scan + ordered expansion (HMJ_EXPAND_FK_FANOUT=1), and the composite sort where it applies.
usage: exp_fk_expand.py [log2 probe rows = 28]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 28
os.environ["HMJ_GTABLE_SORT"] = "0"
os.environ["HMJ_EXPAND_FK_FANOUT"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_EXPAND_FK_FANOUT"] = "1"
ex1 = H.Executor(0)
del os.environ["HMJ_GTABLE_SORT"]
os.environ["HMJ_EXPAND_FK_FANOUT"] = "0"
os.environ["HMJ_GTABLE_SORT_FANOUT"] = "1"
ex2 = H.Executor(0)


def timed(e, R, S, fl, reps=3):
    for _ in range(2):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


for k in range(P - 11, P - 2):
    R, S = ex0.gen_build(1 << k), ex0.gen_uniform_domain(1 << P, 1 << k)
    fl = H.HMJ_ORDERED | H.HMJ_CHECKSUM
    out = []
    cks = []
    for e, name in ((ex0, "one-pass write"), (ex1, "expansion"), (ex2, "composite sort")):
        e.set_profiling(True)
        m, r = timed(e, R, S, fl)
        t = e.last_timing()
        cks.append(r.checks())
        out.append("%s %.3f ms (b%d path %#x; part %.2f count %.2f write %.2f)" % (name, m, t["radix_bits"], t["path"], t["ms_partition_build"] + t["ms_partition_probe"], t["ms_probe_count"], t["ms_probe_write"]))
        e.release_result()
    print("nb=2^%d np=2^%d fan-out %d | %s%s" % (k, P, 1 << (P - k), " | ".join(out), "" if cks[0] == cks[1] == cks[2] else " MISMATCH"), flush=True)
    del R, S
