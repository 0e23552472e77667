#!/usr/bin/env python3
"""Host-resident entry point across sizes that are not powers of two: ms per call (warm) and GB/s of PCIe traffic
(16 B per row up, 24 B per result row down in ordered mode), against what one hipMemcpy reaches (~55 GB/s)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hashmergejoin_amd as H

ex = H.Executor(0)
for k in range(16, 27):
    for mul in (1.0, 1.5):
        n = int((1 << k) * mul) + 3
        R, S = ex.gen_build(n), ex.gen_probe(n, n)
        Bh, Ph = R.cpu().numpy().view(np.uint64), S.cpu().numpy().view(np.uint64)
        del R, S
        row = []
        for fl, name in [(0, "count"), (H.HMJ_ORDERED, "ord")]:
            for _ in range(2):
                ex.join_host(Bh, Ph, fl)
            reps = 5 if k < 24 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                r = ex.join_host(Bh, Ph, fl)
            ms = (time.perf_counter() - t0) / reps * 1e3
            assert int(r.n_matches) == n
            nbytes = 32 * n + (24 * n if fl else 0)
            row.append("%s %.3f ms %.1f GB/s" % (name, ms, nbytes / ms / 1e6))
        print("n=%9d | %s" % (n, " | ".join(row)), flush=True)
        ex.release_result()
