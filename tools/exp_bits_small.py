#!/usr/bin/env python3
"""Count-mode joins of 2^24 ... 2^27 rows per side: the planner's own radix bits against forced totals around it (ms per join,
mean of 20 back-to-back calls; path and pass bits printed).  usage: exp_bits_small.py [log2 sizes, comma separated]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [24, 25, 26, 27]
ex = H.Executor(0)
for lg in sizes:
    n = 1 << lg
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    row = []
    for bits in [None, lg - 13, lg - 12, lg - 11, lg - 10]:
        ex.set_radix_bits(bits)
        for rnd in range(2):
            for _ in range(3):
                r = ex.join_device(R, S, 0)
            assert int(r.n_matches) == n
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ex.join_device(R, S, 0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
        t = ex.last_timing()
        row.append("%s: %.3f ms (bits %d in %d passes, path 0x%x)" % ("planner" if bits is None else "forced %d" % bits, ms, t["radix_bits"], t["radix_passes"], t["path"]))
    ex.set_radix_bits(None)
    print("2^%d x 2^%d count | %s" % (lg, lg, " | ".join(row)), flush=True)
    del R, S
ex.close()
