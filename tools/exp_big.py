#!/usr/bin/env python3
"""Joins beyond 2^28 * 1.06 rows (the reference's own benchmark sweeps to 10^9, hashjoin_bench.cc:269-283): count mode, slab
path with 9-bit passes against the exact path (HMJ_SLAB=0), results checked against the generator's closed forms.
usage: exp_big.py [sizes in millions of rows ...]   (default: 268 285 349 456 500 537 800 1000 1074)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

M64 = (1 << 64) - 1
VAL_XOR = 0x9E3779B97F4A7C15


def sum_xor_range(n, c):  # sum of (j ^ c) for j < n, mod 2^64
    total = 0
    for b in range(64):
        period = 1 << (b + 1)
        full, rem = divmod(n, period)
        ones = full * (1 << b) + max(0, rem - (1 << b))
        if (c >> b) & 1:
            ones = n - ones
        total += ones << b
    return total & M64


sizes = [int(float(x) * 1e6) for x in sys.argv[1:]] or [268435459, 285000000, 348966095, 456000000, 500000000, 1 << 29, 800000000, 1000000000, 1 << 30]
os.environ.pop("HMJ_SLAB", None)
ex = H.Executor(0)
ex0 = None
if os.environ.get("EXP_BIG_NO_EXACT") != "1":  # (2^30 x 2^30: the two contexts' workspaces do not fit the card side by side)
    os.environ["HMJ_SLAB"] = "0"
    ex0 = H.Executor(0)
    os.environ.pop("HMJ_SLAB", None)
for n in sizes:
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    row = []
    for e, name in ((ex, "slab"), (ex0, "exact")):
        if e is None:
            continue
        e.set_profiling(False)
        for _ in range(2):
            r = e.join_device(R, S, 0)
        assert int(r.n_matches) == n and int(r.sum_r) == (n * (n - 1) // 2) & M64 and int(r.sum_s) == sum_xor_range(n, VAL_XOR), (name, n)
        torch.cuda.synchronize()
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            e.join_device(R, S, 0)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        e.set_profiling(True)
        e.join_device(R, S, 0)
        t = e.last_timing()
        row.append("%s %.3f ms %.4f ns/row path %#x b%d A %.3f B %.3f probe %.3f" % (
            name, wall, wall * 1e6 / n, t["path"], t["radix_bits"], t["ms_scatter_pass0"] / 2, t["ms_scatter_pass1"] / 2, t["ms_probe_count"]))
    print("n=%11d | %s" % (n, " | ".join(row)), flush=True)
    del R, S
    if n >= 800000000:  # let the two contexts' workspaces go before the next size (2 x 100 GB at 2^30)
        pass
ex.close()
if ex0 is not None:
    ex0.close()
