#!/usr/bin/env python3
"""One-shot (first call of a fresh context) and warm time of the host entry point at 2^26 rows, ordered rows back --
what HashMergeJoin(r.begin(), ...) costs a caller once, and in a loop.  Run with HMJ_PLACE=0 and =4."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hashmergejoin_amd as H

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << log2n
g = H.Executor(0)
B = g.gen_build(n).cpu().numpy().view(np.uint64)
P = g.gen_probe(n, n).cpu().numpy().view(np.uint64)
g.close()
torch.cuda.empty_cache()
ex = H.Executor(0)
ts = []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ex.join_host(B, P, H.HMJ_ORDERED)
    ts.append((time.perf_counter() - t0) * 1e3)
    assert int(r.n_matches) == n
print("HMJ_PLACE=%s 2^%d host entry, ordered: one-shot %.1f ms, then %.1f / %.1f / %.1f ms; placement %s" % (
    os.environ.get("HMJ_PLACE", "default"), log2n, ts[0], ts[1], ts[2], ts[3], [(b["name"], b["candidates"], b["ms_search"]) for b in ex.placement_info()]), flush=True)
