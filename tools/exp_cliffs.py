#!/usr/bin/env python3
"""Join time per row across sizes that are NOT powers of two (n = 1.0, 1.3, 1.7 x 2^k) and the three main modes, to find
plan cliffs: a size whose ns per row is far above its neighbours' has fallen off a fast path.  Prints wall time per call
(profiling off), the path bits and the plan."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

lo = int(sys.argv[1]) if len(sys.argv) > 1 else 14
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 27
ex = H.Executor(0)
for k in range(lo, hi + 1):
    for mul in (1.0, 1.3, 1.7):
        n = int((1 << k) * mul) + 3
        R, S = ex.gen_build(n), ex.gen_probe(n, n)
        row = []
        for fl, name in [(0, "count"), (H.HMJ_MATERIALIZE, "mat"), (H.HMJ_ORDERED, "ord")]:
            ex.set_profiling(False)
            for _ in range(3):
                ex.join_device(R, S, fl)
            torch.cuda.synchronize()
            reps = 10 if k < 24 else 4
            t0 = time.perf_counter()
            for _ in range(reps):
                ex.join_device(R, S, fl)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            ex.set_profiling(True)
            ex.join_device(R, S, fl)
            t = ex.last_timing()
            row.append("%s %.3f ms %.2f ns/row %#x b%d" % (name, wall, wall * 1e6 / n, t["path"], t["radix_bits"]))
        print("n=%10d | %s" % (n, " | ".join(row)), flush=True)
        del R, S
        ex.release_result()
