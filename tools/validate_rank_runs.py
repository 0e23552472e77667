#!/usr/bin/env python3
"""Randomized check of ordered joins of a small build side (unique keys) under a long probe side -- the rank-run form in all its shapes
(whole runs, runs cut by payload position, several ranks to a partition, 256 / 512 / 1024-thread sorts) and its fallbacks -- against an
independent torch implementation: exact row sequence (key, rval, sval ascending, unsigned).
Every iteration draws sizes, payload kind, missing build rows and the switches HMJ_RANK_RUNS_MAX_CUT / _MAX_LEVEL / _MAX_GROUP.
Usage: tools/validate_rank_runs.py [iters] [seed] [max log2 probe rows = 24]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hashmergejoin_amd as H
from hashmergejoin_amd.join import _memcpy_d2d

dev = torch.device("cuda:0")
SIGN = torch.tensor(-(1 << 63), dtype=torch.int64, device=dev)


def reference(B, P):
    """rows of the join ordered by (key, sval) unsigned; build keys are unique"""
    kb = B[:, 0] ^ SIGN
    sk, order = torch.sort(kb)
    pos = torch.searchsorted(sk, P[:, 0] ^ SIGN).clamp_(max=sk.numel() - 1)
    hit = sk[pos] == (P[:, 0] ^ SIGN)
    rank = pos[hit]
    sv = P[:, 1][hit]
    o1 = torch.argsort(sv ^ SIGN, stable=True)
    o2 = torch.argsort(rank[o1], stable=True)
    idx = o1[o2]
    r = rank[idx]
    return sk[r] ^ SIGN, B[:, 1][order][r], sv[idx]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    maxlog = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    cpu = torch.Generator()
    cpu.manual_seed(seed)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi, (1,), generator=cpu).item())
    bad = 0
    paths = {}
    for it in range(iters):
        env = {"HMJ_GTABLE_SORT_FANOUT": "1", "HMJ_RANK_RUNS_MAX_CUT": str([0, 2, 10][ri(0, 3)]), "HMJ_RANK_RUNS_MAX_LEVEL": str(ri(0, 3)),
               "HMJ_RANK_RUNS_MAX_GROUP": str([0, 3][ri(0, 2)])}
        os.environ.update(env)
        ex = H.Executor(0)
        for k in env:
            del os.environ[k]
        lp = ri(16, maxlog)
        npb = (1 << lp) + ri(0, 1 << (lp - 1)) - ri(0, 300)
        big_build = ri(0, 3) == 0 and npb >= (1 << 23)  # more than 2^18 build rows: ranks grouped
        nb = ri(262145, min(npb // 16, 1 << 20) + 1) if big_build else ri(4, max(5, min(npb // 16, 1 << 18)))
        if ri(0, 3) == 0:
            nb = min(nb, ri(4, 3000))  # long runs
        keys = torch.randperm(1 << 22, device=dev, generator=g)[:nb].to(torch.int64) * 0x9E3779B97F4A7C15 if nb <= (1 << 22) else None
        B = torch.stack([keys, torch.arange(nb, device=dev, dtype=torch.int64) * 3 + 1], 1).contiguous()
        pk = keys[torch.randint(0, nb, (npb,), device=dev, generator=g)]
        miss = ri(0, 4) == 0
        if miss:
            m = torch.randint(0, 7, (npb,), device=dev, generator=g) == 0
            pk = torch.where(m, pk + 1, pk)
        kind = ["rowid", "rowid_desc", "random", "wide", "offset", "clustered", "ties"][ri(0, 7)]
        ar = torch.arange(npb, device=dev, dtype=torch.int64)
        if kind == "rowid":
            sv = ar * ri(1, 9) + ri(0, 1 << 40)
        elif kind == "rowid_desc":
            sv = (npb - ar) * ri(1, 9)
        elif kind == "random":
            sv = torch.randperm(npb, device=dev, generator=g).to(torch.int64)
        elif kind == "wide":
            sv = torch.randint(-(1 << 63), (1 << 63) - 1, (npb,), device=dev, generator=g, dtype=torch.int64)
        elif kind == "offset":
            sv = torch.randint(0, 1 << 30, (npb,), device=dev, generator=g, dtype=torch.int64) - (1 << 62)
        elif kind == "clustered":  # most payloads in a narrow band, a few far away
            sv = torch.randint(0, 1 << 20, (npb,), device=dev, generator=g, dtype=torch.int64)
            sv[:: ri(50, 5000)] = 1 << 50
        else:
            sv = torch.randint(0, ri(2, 40), (npb,), device=dev, generator=g, dtype=torch.int64)
        P = torch.stack([pk, sv], 1).contiguous()
        wk, wr, ws = reference(B, P)
        for fl in (H.HMJ_ORDERED, H.HMJ_ORDERED | H.HMJ_CHECKSUM):
            r = ex.join_device(B, P, fl)
            t = ex.last_timing()
            n = int(r.n_matches)
            cols = []
            for ptr in (r.key, r.rval, r.sval):
                c = torch.empty(n, dtype=torch.int64, device=dev)
                if n:
                    _memcpy_d2d(torch, c, ptr, n * 8)
                cols.append(c)
            ok = n == wk.numel() and all(bool(torch.equal(a, b)) for a, b in zip(cols, (wk, wr, ws)))
            key = "%#x" % (t["path"] & (H.HMJ_PATH_RANK_RUNS | H._lib.HMJ_PATH_RANK_LOOKUP_IN_PASS | H.HMJ_PATH_ORDER_BY_RANK_SORT))
            paths[key] = paths.get(key, 0) + 1
            if not ok:
                bad += 1
                print("MISMATCH it=%d nb=%d np=%d kind=%s miss=%s env=%s fl=%#x path=%#x n=%d want=%d" % (it, nb, npb, kind, miss, env, fl, t["path"], n, wk.numel()), flush=True)
        print("it=%d nb=%d np=%d f=%.0f kind=%s miss=%s cut=%s lvl=%s grp=%s path=%#x bits=%d" % (
            it, nb, npb, npb / nb, kind, miss, env["HMJ_RANK_RUNS_MAX_CUT"], env["HMJ_RANK_RUNS_MAX_LEVEL"], env["HMJ_RANK_RUNS_MAX_GROUP"], t["path"], t["radix_bits"]), flush=True)
        ex.release_result()
        ex.close()
        del B, P, wk, wr, ws, cols
    print("paths:", paths)
    print("RESULT: %d mismatches in %d iterations" % (bad, iters))
    sys.exit(1 if bad else 0)


main()
