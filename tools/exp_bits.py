#!/usr/bin/env python3
"""Whole-join time at 2^log2n for forced total radix bits (planner default vs alternatives)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ex = H.Executor(0); ex.set_profiling(True)
n = 1 << log2n
R, S = ex.gen_build(n), ex.gen_probe(n, n)
for bits in [int(x) for x in sys.argv[2:]] or [16, 17, 18]:
    ex.set_radix_bits(bits)
    best = None
    for _ in range(4):
        r = ex.join_device(R, S, 0)
        assert int(r.n_matches) == n or os.environ.get('HMJ_DEBUG_ABLATE')
        t = ex.last_timing()
        if best is None or t["ms_total"] < best["ms_total"]:
            best = t
    print("B=%d total %.3f ms  hist %.3f scatter %.3f probe %.3f" % (bits, best["ms_total"], best["ms_hist"], best["ms_scatter"], best["ms_probe_count"]), flush=True)
