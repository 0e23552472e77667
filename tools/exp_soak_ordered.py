#!/usr/bin/env python3
"""Soak of the one-pass ordered write: hundreds of ordered joins (unique keys with and without unmatched probe rows ->
chained and unchained output offsets; foreign-key joins of several fan-outs) on one executor.  Every join's match count
is checked against its closed form, no join may report a look-back timeout, and the share of joins that stayed on the
one-pass path is printed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import hashmergejoin_amd as H

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ex = H.Executor(0)
ex.set_profiling(True)
g = torch.Generator()
g.manual_seed(11)
one_pass = timeouts = 0
t0 = time.time()
for it in range(iters):
    kind = int(torch.randint(0, 3, (1,), generator=g))
    lb = int(torch.randint(20, 26, (1,), generator=g))
    if kind == 0:    # unique keys, every probe row matches
        n = (1 << lb) + int(torch.randint(0, 5000, (1,), generator=g))
        R, S, want = ex.gen_build(n), ex.gen_probe(n, n), n
    elif kind == 1:  # unique keys, one probe row in miss_mod unmatched
        n = (1 << lb) + int(torch.randint(0, 5000, (1,), generator=g))
        mm = int(torch.randint(2, 9, (1,), generator=g))
        R, S = ex.gen_build(n), ex.gen_probe(n, n, miss_mod=mm)
        want = n - (n + mm - 1) // mm
    else:            # foreign-key join, fan-out 2 ... 128
        f = 1 << int(torch.randint(1, 8, (1,), generator=g))
        nb = max(1 << 12, (1 << lb) // f)
        R, S, want = ex.gen_build(nb), ex.gen_uniform_domain(nb * f, nb), nb * f
    r = ex.join_device(R, S, H.HMJ_ORDERED)
    t = ex.last_timing()
    assert int(r.n_matches) == want, (it, kind, lb, int(r.n_matches), want)
    one_pass += bool(t["path"] & H.HMJ_PATH_SORTED_WRITE)
    timeouts += bool(t["path"] & H.HMJ_PATH_LOOKBACK_TIMEOUT)
    if it % 50 == 0:
        print("iter %d: one-pass %d / %d, look-back timeouts %d, %.1f s" % (it, one_pass, it + 1, timeouts, time.time() - t0), flush=True)
    del R, S
    ex.release_result()
print("done: %d joins, one-pass %d, look-back timeouts %d" % (iters, one_pass, timeouts))
assert timeouts == 0
