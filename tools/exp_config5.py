#!/usr/bin/env python3
"""BASELINE configs[4]: Zipf(0.9) build side 2^24 (duplicate keys), uniform probe 2^30 over the same domain.
Per-phase time of the cross-product count and of the first-wins sum."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hashmergejoin_amd as H
log2b = int(sys.argv[1]) if len(sys.argv) > 1 else 24
log2p = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M64 = (1 << 64) - 1
dom = 1 << log2b
w = 1.0 / np.arange(1, dom + 1, dtype=np.float64) ** 0.9
cdf = np.cumsum(w) / w.sum()
thr = np.empty(dom, np.uint64)
big = cdf >= 1.0 - 2.0 ** -53
thr[~big] = (cdf[~big] * 2.0 ** 64).astype(np.uint64)
thr[big] = np.uint64(M64); thr[-1] = np.uint64(M64)
ex = H.Executor(0); ex.set_profiling(True)
R = ex.gen_from_cdf(1 << log2b, torch.from_numpy(thr.view(np.int64).copy()).cuda())
S = ex.gen_uniform_domain(1 << log2p, dom)
for name, fl, (b, p) in [("cross-product count", 0, (R, S)), ("first_wins+sum_probe", H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE, (R, S)),
                         ("swapped (skewed probe)", 0, (S, R))]:
    best = None
    for _ in range(3):
        r = ex.join_device(b, p, fl)
        t = ex.last_timing()
        if best is None or t["ms_total"] < best["ms_total"]: best = t
    print("%-24s n_matches %d total %.3f ms | part_build %.3f part_probe %.3f probe %.3f -> %.2f G probe tuples/s" % (
        name, int(r.n_matches), best["ms_total"], best["ms_partition_build"], best["ms_partition_probe"], best["ms_probe_count"],
        p.shape[0] / best["ms_total"] / 1e6), flush=True)
