#!/usr/bin/env python3
"""Small joins (the low end of the reference's sweep, hashjoin_bench.cc:269-279: 10^4 ... 10^6 rows per side): device-resident latency per
call in count / materialise / ordered mode, and the host-resident call (hmj_join_u64: H2D + join + D2H of the result rows)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hashmergejoin_amd as H

ex = H.Executor(0)
for n in [10_000, 50_000, 100_000, 500_000, 1_000_000, 5_000_000]:
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    Rh, Sh = R.cpu().numpy().view(np.uint64), S.cpu().numpy().view(np.uint64)
    row = []
    for fl, name in [(H.HMJ_CHECKSUM, "count"), (H.HMJ_MATERIALIZE, "mat"), (H.HMJ_ORDERED, "ord")]:
        for _ in range(5):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        row.append("%s %7.1f us (%#x)" % (name, us, ex.last_timing()["path"]))
    for _ in range(3):
        ex.join_host(Rh, Sh, H.HMJ_ORDERED)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ex.join_host(Rh, Sh, H.HMJ_ORDERED)
    us = (time.perf_counter() - t0) / reps * 1e6
    row.append("host ord %8.1f us" % us)
    print("n=%8d | %s" % (n, " | ".join(row)), flush=True)
