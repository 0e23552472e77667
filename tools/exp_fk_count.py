#!/usr/bin/env python3
"""Count-mode FK joins: default plan (bits from the build side) vs bits from the probe side."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
ex = H.Executor(0); ex.set_profiling(True)
for log2b, log2p in [(24, 26), (24, 27), (23, 26), (25, 26), (25, 27), (25, 28), (26, 28), (27, 28), (24, 28)]:
    R = ex.gen_build(1 << log2b)
    S = ex.gen_uniform_domain(1 << log2p, 1 << log2b)
    for bits in [None, H.plan(1 << log2p)[0]]:
        ex.set_radix_bits(bits)
        best = None
        for _ in range(3):
            r = ex.join_device(R, S, 0); t = ex.last_timing()
            if best is None or t["ms_total"] < best["ms_total"]: best = t
        assert int(r.n_matches) == 1 << log2p
        print("build 2^%d probe 2^%d bits %2d: total %.2f ms | part_build %.2f part_probe %.2f probe %.2f hist %.2f" % (
            log2b, log2p, best["radix_bits"], best["ms_total"], best["ms_partition_build"], best["ms_partition_probe"], best["ms_probe_count"], best["ms_hist"]), flush=True)
    ex.set_radix_bits(None)
