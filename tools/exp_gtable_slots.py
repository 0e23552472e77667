#!/usr/bin/env python3
"""Global table, count mode: ms per join over table load factors (slots per build row) -- developer build only
(HMJ_LIB=build/variants/libhmj_dev.so: HMJ_GTABLE_SLOTS is compiled out of the release library).
usage: exp_gtable_slots.py [P=26] [kmin=13] [kmax=18]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 26
kmin = int(sys.argv[2]) if len(sys.argv) > 2 else 13
kmax = int(sys.argv[3]) if len(sys.argv) > 3 else 18
os.environ["HMJ_GTABLE_MAX_LOG2"] = "26"
exs = {}
for slots in (2, 3, 4, 8, 16):
    os.environ["HMJ_GTABLE_SLOTS"] = str(slots)
    exs[slots] = H.Executor(0)
n_p = 1 << P
for k in range(kmin, kmax + 1):
    nb = 1 << k
    R = exs[2].gen_build(nb)
    for name, S in (("fk", exs[2].gen_uniform_domain(n_p, nb)), ("miss3", exs[2].gen_probe(n_p, nb, miss_mod=3))):
        row = []
        for slots, e in exs.items():
            for _ in range(2):
                r = e.join_device(R, S, 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                r = e.join_device(R, S, 0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 6 * 1e3
            row.append("slots%-2d %.3f%s" % (slots, ms, "" if e.last_timing()["path"] & H.HMJ_PATH_GLOBAL_TABLE else "(!gt)"))
        print("nb=2^%d np=2^%d %-5s | %s" % (k, P, name, "  ".join(row)), flush=True)
