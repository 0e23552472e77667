#!/usr/bin/env python3
"""Materialising joins with duplicate build keys at 2^k rows a side (every 4th build row repeats the key of the row before it,
the probe side hits every build key once): phases of the unordered and the ordered result, with the ordered expansion
(probe_expand_ordered_kernel) and without it (HMJ_ORDERED_EXPANSION=0: write in probe order, sort the rows).
usage: exp_dup_build.py [log2 rows = 28]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

lg = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << lg
os.environ["HMJ_ORDERED_EXPANSION"] = "0"
ex0 = H.Executor(0)
del os.environ["HMJ_ORDERED_EXPANSION"]
ex1 = H.Executor(0)
R, S = ex1.gen_build(n), ex1.gen_probe(n, n)
R[1::4, 0] = R[0::4, 0]
keep = ("ms_total", "ms_partition_build", "ms_partition_probe", "ms_probe_count", "ms_probe_write", "ms_order")
for fl, name in ((H.HMJ_MATERIALIZE, "rows"), (H.HMJ_ORDERED, "ordered")):
    res = []
    for e in (ex0, ex1):
        e.set_profiling(True)
        for _ in range(3):
            r = e.join_device(R, S, fl)
        t = e.last_timing()
        res.append((int(r.n_matches), int(r.sum_r), int(r.sum_s)))
        print("2^%d x 2^%d, every 4th build key doubled, %s, %s: %s path %#x bits %d" % (
            lg, lg, name, "write + sort" if e is ex0 else "ordered expansion", {k: round(t[k], 3) for k in keep if t[k] > 0}, t["path"], t["radix_bits"]), flush=True)
        e.release_result()
    assert res[0] == res[1], res
