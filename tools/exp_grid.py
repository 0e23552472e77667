#!/usr/bin/env python3
"""Build x probe size grid, foreign-key joins (uniform over unique build keys): ordered / materialise / count, ms per join with path and bits.
A cell far above its neighbours has fallen between the paths' gates.  usage: exp_grid.py log2_probe [log2_build_lo log2_build_hi]"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

lp = int(sys.argv[1]) if len(sys.argv) > 1 else 28
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hi = int(sys.argv[3]) if len(sys.argv) > 3 else min(lp, 28)
ex = H.Executor(0)
npb = 1 << lp
for lb in range(lo, hi + 1):
    for mul in (1.0, 1.5):
        nb = int((1 << lb) * mul) + (3 if mul != 1.0 else 0)
        if nb > npb:
            continue
        R, S = ex.gen_build(nb), ex.gen_uniform_domain(npb, nb)
        row = []
        for fl, name in [(H.HMJ_ORDERED, "ord"), (H.HMJ_MATERIALIZE, "mat"), (H.HMJ_CHECKSUM, "count")]:
            for _ in range(2):
                ex.join_device(R, S, fl)
            torch.cuda.synchronize()
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ex.join_device(R, S, fl)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            t = ex.last_timing()
            row.append("%s %8.3f ms %#9x b%-2d" % (name, wall, t["path"], t["radix_bits"]))
        print("np=2^%d nb=%10d (2^%4.1f) | %s" % (lp, nb, math.log2(nb), " | ".join(row)), flush=True)
        ex.release_result()
        del R, S
