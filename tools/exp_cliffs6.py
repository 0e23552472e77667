#!/usr/bin/env python3
"""Sixth cliff sweep (round 4: rows placed behind a result cursor -- global table, one-pass slab walk): UNORDERED materialising joins and
their first-wins form over a grid of build x probe sizes, ns per (build + probe) row with path and bits.  A cell far above its neighbours
has fallen between the paths' gates."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

ex = H.Executor(0)
probes = [int(x) for x in sys.argv[1:]] or [16, 18, 20, 22, 24, 26, 27]
for lp in probes:
    npb = (1 << lp) + 11
    for lb in range(10, 25):
        for mul in (1.0, 1.5):
            nb = int((1 << lb) * mul) + 3
            if nb > 4 * npb:
                continue
            R = ex.gen_build(nb)
            S = ex.gen_uniform_domain(npb, nb) if npb >= nb else ex.gen_probe(npb, nb)
            row = []
            for fl, name in [(H.HMJ_MATERIALIZE, "rows"), (H.HMJ_MATERIALIZE | H.HMJ_FIRST_WINS | H.HMJ_CHECKSUM, "rows first+ck")]:
                ex.set_profiling(False)
                for _ in range(3):
                    ex.join_device(R, S, fl)
                torch.cuda.synchronize()
                reps = 8 if lp <= 24 else 4
                t0 = time.perf_counter()
                for _ in range(reps):
                    ex.join_device(R, S, fl)
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / reps * 1e3
                t = ex.last_timing()
                row.append("%s %7.3f ms %6.3f ns/row %#7x b%-2d" % (name, wall, wall * 1e6 / (nb + npb), t["path"], t["radix_bits"]))
            print("np=2^%d nb=%9d (2^%.1f) | %s" % (lp, nb, __import__("math").log2(nb), " | ".join(row)), flush=True)
            del R, S
