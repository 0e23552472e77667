#!/usr/bin/env python3
"""Small build sides: the global-table path (gtable.hip) against the partitioned paths, count modes.
For each build size 2^k x probe 2^P rows: ms per join with HMJ_GTABLE=0 and =1 (results compared), several probe grids.
usage: exp_gtable.py [P=26] [kmin=12] [kmax=22]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 26
kmin = int(sys.argv[2]) if len(sys.argv) > 2 else 12
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 22
os.environ["HMJ_GTABLE"] = "0"
ex0 = H.Executor(0)
os.environ["HMJ_GTABLE"] = "1"
os.environ["HMJ_GTABLE_MAX_LOG2"] = "26"
os.environ["HMJ_GTABLE_FANOUT"] = "1"
exs = {}
for wg in (4, 8, 16):
    os.environ["HMJ_GTABLE_WG"] = str(wg)
    exs[wg] = H.Executor(0)


def timed(e, R, S, fl, reps=6):
    for _ in range(2):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = e.join_device(R, S, fl)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


n_p = 1 << P
for k in range(kmin, kmax + 1):
    nb = 1 << k
    R = ex0.gen_build(nb)
    for name, S in (("fk", ex0.gen_uniform_domain(n_p, nb)), ("miss3", ex0.gen_probe(n_p, nb, miss_mod=3))):
        for fl, fname in ((0, "count"), (H.HMJ_CHECKSUM | H.HMJ_SUM_PROBE, "checks"), (H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE, "first")):
            ms0, r0 = timed(ex0, R, S, fl)
            assert not ex0.last_timing()["path"] & H.HMJ_PATH_GLOBAL_TABLE
            want = r0.checks() if fl & H.HMJ_CHECKSUM else (int(r0.n_matches), int(r0.sum_r), int(r0.sum_s), int(r0.sum_probe_all))
            row = []
            for wg, e in exs.items():
                ms, r = timed(e, R, S, fl)
                got = r.checks() if fl & H.HMJ_CHECKSUM else (int(r.n_matches), int(r.sum_r), int(r.sum_s), int(r.sum_probe_all))
                ok = got == want and e.last_timing()["path"] & H.HMJ_PATH_GLOBAL_TABLE
                row.append("wg%d %.3f%s" % (wg, ms, "" if ok else " MISMATCH/path %#x" % e.last_timing()["path"]))
            print("nb=2^%d np=2^%d %-5s %-6s | partitioned %.3f ms (b%d path %#x) | gtable %s" % (
                k, P, name, fname, ms0, ex0.last_timing()["radix_bits"], ex0.last_timing()["path"], "  ".join(row)), flush=True)
        del S
    del R
