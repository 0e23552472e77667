#!/usr/bin/env python3
"""Host-resident entry point (hmj_join_u64): first call vs warm call, staged multi-thread upload vs one hipMemcpy."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hashmergejoin_amd as H
from oracle.pyoracle import Oracle
o = Oracle()
for log2n in [22, 24, 26]:
    n = 1 << log2n
    B, P = o.gen_build(n), o.gen_probe(n, n)
    for threads in [1, 8]:
        for fl, name in [(0, "count"), (H.HMJ_ORDERED, "ordered rows")]:
            ex = H.Executor(0)
            ex.L.hmj_set_host_threads(ex.h, threads)
            ts = []
            for rep in range(3):
                t0 = time.perf_counter(); r = ex.join_host(B, P, fl); ts.append((time.perf_counter() - t0) * 1e3)
            assert int(r.n_matches) == n
            ex.set_profiling(True); ex.join_host(B, P, fl); t = ex.last_timing()
            print("2^%d %-12s host_threads=%d: first %.1f ms, then %.1f / %.1f ms  (h2d %.1f d2h %.1f gpu %.1f)" % (
                log2n, name, threads, ts[0], ts[1], ts[2], t["ms_h2d"], t["ms_d2h"], t["ms_total"] - t["ms_h2d"] - t["ms_d2h"]), flush=True)
            ex.close()
