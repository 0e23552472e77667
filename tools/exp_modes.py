#!/usr/bin/env python3
"""Per-phase time of the join in every output mode."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ex = H.Executor(0); ex.set_profiling(True)
n = 1 << log2n
R, S = ex.gen_build(n), ex.gen_probe(n, n)
for name, fl in [("count", 0), ("count+checksum", H.HMJ_CHECKSUM), ("first_wins+sum_probe", H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE),
                 ("materialize", H.HMJ_MATERIALIZE), ("ordered", H.HMJ_ORDERED)]:
    best = None
    for _ in range(3):
        r = ex.join_device(R, S, fl)
        t = ex.last_timing()
        if best is None or t["ms_total"] < best["ms_total"]:
            best = t
    print("%-22s total %.3f | part %.3f probe_count %.3f out_scan %.3f probe_write %.3f order %.3f" % (
        name, best["ms_total"], best["ms_hist"] + best["ms_scatter"] + best["ms_scan"], best["ms_probe_count"], best["ms_out_scan"],
        best["ms_probe_write"], best["ms_order"]), flush=True)
