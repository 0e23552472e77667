#!/usr/bin/env python3
"""Ordered joins of a small build side (rank forms): the build side sorted by one MSD pass + an LDS sort per partition against
eight LSD passes (HMJ_BUILD_SORT_MSD=0; developer build: tools/build_variant.sh dev, HMJ_LIB=build/variants/libhmj_dev.so).
Both contexts must return the same checksums; ms per join (wall, mean of 8).  usage: exp_build_sort.py [log2 probe rows = 26]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

P = int(sys.argv[1]) if len(sys.argv) > 1 else 26
exs = {}
for m in (0, 1):
    os.environ["HMJ_BUILD_SORT_MSD"] = str(m)
    exs[m] = H.Executor(0)
bad = 0
for k in (4, 8, 10, 11, 12, 14, 16, 17, 18, 19, 20):
    nb, np_ = 1 << k, 1 << P
    if np_ < 16 * nb:
        continue
    R = exs[0].gen_build(nb)
    S = exs[0].gen_uniform_domain(np_, nb)
    row, want = [], None
    for rnd in range(2):
        for m, e in exs.items():
            for _ in range(2):
                r = e.join_device(R, S, H.HMJ_ORDERED | H.HMJ_CHECKSUM)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                r = e.join_device(R, S, H.HMJ_ORDERED | H.HMJ_CHECKSUM)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 8 * 1e3
            got = r.checks()
            want = want or got
            ok = got == want
            bad += not ok
            row.append("%s %.3f%s" % ("msd" if m else "lsd", ms, "" if ok else " MISMATCH"))
    print("2^%-2d x 2^%d ordered | %s | path 0x%x" % (k, P, "  ".join(row), exs[1].last_timing()["path"]), flush=True)
    e.release_result()
    exs[0].release_result()
print("RESULT: %d mismatches" % bad)
