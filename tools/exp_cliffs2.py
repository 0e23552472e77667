#!/usr/bin/env python3
"""Second cliff sweep: asymmetric sizes, unmatched probe rows, dense / sorted / shifted integer keys, duplicate build keys,
in count, materialise and ordered mode.  Prints ms per call and ns per (build + probe) row; a line far above its
neighbours has fallen off a fast path (the path bits say which one it took)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

ex = H.Executor(0)
dev = torch.device("cuda", 0)
MODES = [(0, "count"), (H.HMJ_MATERIALIZE, "mat"), (H.HMJ_ORDERED, "ord")]


def run(tag, R, S, modes=MODES):
    nb, npb = R.shape[0], S.shape[0]
    row = []
    for fl, name in modes:
        ex.set_profiling(False)
        for _ in range(3):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        ex.set_profiling(True)
        r = ex.join_device(R, S, fl)
        t = ex.last_timing()
        row.append("%s %.3f ms %.3f ns/row %#x b%d m=%d" % (name, wall, wall * 1e6 / (nb + npb), t["path"], t["radix_bits"], int(r.n_matches)))
    print("%-34s nb=%9d np=%9d | %s" % (tag, nb, npb, " | ".join(row)), flush=True)
    ex.release_result()


def rows(keys, vals=None):
    keys = keys.to(torch.int64)
    if vals is None:
        vals = torch.arange(keys.numel(), device=dev, dtype=torch.int64)
    return torch.stack([keys, vals], 1).contiguous()


which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "asym"):
    for lb, lp in [(26, 20), (26, 22), (26, 24), (24, 26), (22, 26), (20, 26), (16, 26), (12, 26), (26, 12), (27, 23), (23, 27)]:
        nb, npb = (1 << lb) + 11, (1 << lp) + 7
        R = ex.gen_build(nb)
        S = ex.gen_probe(npb, nb) if npb <= nb else ex.gen_uniform_domain(npb, nb)
        run("asym 2^%d x 2^%d" % (lb, lp), R, S)
        del R, S
if which in ("all", "miss"):
    n = (1 << 25) + 5
    for mm in (0, 2, 1):  # miss_mod: 0 = none, 2 = every second probe row unmatched, 1 = all unmatched
        R, S = ex.gen_build(n), ex.gen_probe(n, n, miss_mod=mm)
        run("miss_mod %d" % mm, R, S)
        del R, S
if which in ("all", "keys"):
    n = (1 << 25) + 5
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    perm = torch.randperm(n, device=dev, generator=g)
    ar = torch.arange(n, device=dev, dtype=torch.int64)
    for tag, kb, kp in [("dense 0..n shuffled", perm, torch.randperm(n, device=dev, generator=g)),
                        ("dense 0..n sorted both", ar, ar),
                        ("dense + 2^40 offset", perm + (1 << 40), torch.randperm(n, device=dev, generator=g) + (1 << 40)),
                        ("multiples of 4096", perm * 4096, torch.randperm(n, device=dev, generator=g) * 4096),
                        ("top bits only (<< 38)", perm << 38, torch.randperm(n, device=dev, generator=g) << 38)]:
        run(tag, rows(kb), rows(kp))
    del perm, ar
if which in ("all", "dups"):
    n = 1 << 24
    g = torch.Generator(device=dev)
    g.manual_seed(6)
    for d in (2, 8):
        kb = torch.randint(0, n // d, (n,), device=dev, generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62)
        kp = torch.randint(0, n // d, (n,), device=dev, generator=g, dtype=torch.int64) * 0x9E3779B97F4A7C15 % (1 << 62)
        run("dup build keys x%d" % d, rows(kb), rows(kp))
