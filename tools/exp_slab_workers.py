#!/usr/bin/env python3
"""Count-mode joins (slab path) vs the number of pass-A workers, by size.  Needs a -DHMJ_DEV build (HMJ_SLAB_WORKERS is a
developer switch): tools/build_variant.sh dev ; HMJ_LIB=build/variants/libhmj_dev.so python tools/exp_slab_workers.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hashmergejoin_amd as H  # noqa: E402

sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [24, 25, 26, 27, 28]
workers = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [256, 512, 1024, 2048]
ex = H.Executor(0)
for lg in sizes:
    n = 1 << lg
    R, S = ex.gen_build(n), ex.gen_probe(n, n)
    for rnd in range(2):  # two rounds: the order of the settings is not the effect
        for w in workers:
            os.environ["HMJ_SLAB_WORKERS"] = str(w)
            ex.set_profiling(False)
            for _ in range(3):
                r = ex.join_device(R, S, 0)
            assert int(r.n_matches) == n
            torch.cuda.synchronize()
            reps = 20 if lg <= 26 else 8
            t0 = time.perf_counter()
            for _ in range(reps):
                ex.join_device(R, S, 0)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            ex.set_profiling(True)
            ex.join_device(R, S, 0)
            t = ex.last_timing()
            print("2^%d workers %4d round %d: wall %.3f ms  passA %.3f passB %.3f probe %.3f  bits %d path 0x%x" % (
                lg, w, rnd, wall, t.get("ms_scatter_pass0", 0) / 2, t.get("ms_scatter_pass1", 0) / 2, t["ms_probe_count"], t["radix_bits"], t["path"]), flush=True)
    del R, S
ex.close()
