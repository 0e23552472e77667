#!/usr/bin/env python3
"""hmj_sort_u64_device (radix_int_non_inplace / radix_int_inplace replacement) across sizes: G keys/s, uniform and dense keys."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H

ex = H.Executor(0)
dev = torch.device("cuda", 0)
for k in range(12, 29, 2):
    for mul in (1.0, 1.4):
        n = int((1 << k) * mul) + 3
        if n > (1 << 28) + 3:
            continue
        R = ex.gen_build(n)
        D = torch.stack([torch.randperm(n, device=dev), torch.arange(n, device=dev)], 1).contiguous()
        row = []
        for tag, rel in (("uniform", R), ("dense", D)):
            for inplace in (False, True):
                src = rel.clone()
                for _ in range(2):
                    ex.sort_device(src.clone() if inplace else src, inplace=inplace)
                torch.cuda.synchronize()
                reps = 5
                srcs = [src.clone() for _ in range(reps)] if inplace else [src] * reps
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in srcs:
                    ex.sort_device(s, inplace=inplace)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / reps * 1e3
                row.append("%s%s %.3f ms %.2f G/s" % (tag, " inplace" if inplace else "", ms, n / ms / 1e6))
                del srcs, src
        print("n=%10d | %s" % (n, " | ".join(row)), flush=True)
        del R, D
