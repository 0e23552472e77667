#!/usr/bin/env python3
"""Ordered FK joins: unique build keys (2^b), probe side k times larger (uniform over the build keys)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
ex = H.Executor(0); ex.set_profiling(True)
for log2b, log2p in [(24, 24), (24, 25), (24, 26), (22, 26), (24, 28)]:
    R = ex.gen_build(1 << log2b)
    S = ex.gen_uniform_domain(1 << log2p, 1 << log2b)
    for name, fl in [("materialize", H.HMJ_MATERIALIZE), ("ordered", H.HMJ_ORDERED)]:
        best = None
        for _ in range(3):
            r = ex.join_device(R, S, fl); t = ex.last_timing()
            if best is None or t["ms_total"] < best["ms_total"]: best = t
        print("build 2^%d probe 2^%d %-12s total %8.2f ms | part %.2f count %.2f write %.2f order %.2f  (matches %d)" % (
            log2b, log2p, name, best["ms_total"], best["ms_partition_build"] + best["ms_partition_probe"], best["ms_probe_count"],
            best["ms_probe_write"], best["ms_order"], int(r.n_matches)), flush=True)
    ex.release_result()
