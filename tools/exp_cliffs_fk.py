#!/usr/bin/env python3
"""Foreign-key joins of a fixed probe size over fan-outs 1.5 ... 512 (sizes not powers of two), count / materialise /
ordered: ms per call, ns per probe row, path, bits."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

lp = int(sys.argv[1]) if len(sys.argv) > 1 else 27
ex = H.Executor(0)
npb = (1 << lp) + 7
for f in (1.5, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 256, 512):
    nb = int(npb / f) + 3
    R, S = ex.gen_build(nb), ex.gen_uniform_domain(npb, nb)
    row = []
    for fl, name in [(0, "count"), (H.HMJ_MATERIALIZE, "mat"), (H.HMJ_ORDERED, "ord")]:
        ex.set_profiling(False)
        for _ in range(3):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ex.join_device(R, S, fl)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5 * 1e3
        ex.set_profiling(True)
        ex.join_device(R, S, fl)
        t = ex.last_timing()
        row.append("%s %.2f ms %.3f ns %#x b%d" % (name, wall, wall * 1e6 / npb, t["path"], t["radix_bits"]))
    print("f=%5.1f nb=%9d | %s" % (f, nb, " | ".join(row)), flush=True)
    del R, S
    ex.release_result()
