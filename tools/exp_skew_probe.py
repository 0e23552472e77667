#!/usr/bin/env python3
"""Foreign-key shape: unique build keys (2^24), Zipf(theta)-skewed probe side (2^28 rows over those keys)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hashmergejoin_amd as H
log2b = int(sys.argv[1]) if len(sys.argv) > 1 else 24
log2p = int(sys.argv[2]) if len(sys.argv) > 2 else 28
M64 = (1 << 64) - 1
ex = H.Executor(0); ex.set_profiling(True)
dom = 1 << log2b
R = ex.gen_build(dom)
for theta in [0.0, 0.5, 0.9, 1.1]:
    w = 1.0 / np.arange(1, dom + 1, dtype=np.float64) ** theta
    cdf = np.cumsum(w) / w.sum()
    thr = np.empty(dom, np.uint64)
    big = cdf >= 1.0 - 2.0 ** -53
    thr[~big] = (cdf[~big] * 2.0 ** 64).astype(np.uint64)
    thr[big] = np.uint64(M64); thr[-1] = np.uint64(M64)
    S = ex.gen_from_cdf(1 << log2p, torch.from_numpy(thr.view(np.int64).copy()).cuda())
    for name, fl in [("count", 0), ("materialize", H.HMJ_MATERIALIZE)]:
        best = None
        for _ in range(3):
            r = ex.join_device(R, S, fl)
            t = ex.last_timing()
            if best is None or t["ms_total"] < best["ms_total"]: best = t
        assert int(r.n_matches) == (1 << log2p)
        print("theta %.1f %-12s total %.2f ms | part_build %.2f part_probe %.2f probe_count %.2f write %.2f -> %.1f G probe tuples/s" % (
            theta, name, best["ms_total"], best["ms_partition_build"], best["ms_partition_probe"], best["ms_probe_count"], best["ms_probe_write"],
            (1 << log2p) / best["ms_total"] / 1e6), flush=True)
    del S
