#!/usr/bin/env python3
"""Soak: thousands of joins of changing sizes and modes on one executor; device memory in use must level off."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
ex = H.Executor(0)
g = torch.Generator(); g.manual_seed(3)
flags = [0, H.HMJ_CHECKSUM, H.HMJ_MATERIALIZE, H.HMJ_ORDERED, H.HMJ_FIRST_WINS | H.HMJ_SUM_PROBE, H.HMJ_FIRST_WINS | H.HMJ_ORDERED]
free0, total = torch.cuda.mem_get_info()
t0 = time.time()
for it in range(3000):
    n = int(torch.randint(1, 1 << int(torch.randint(8, 23, (1,), generator=g)), (1,), generator=g))
    m = int(torch.randint(1, 1 << int(torch.randint(8, 23, (1,), generator=g)), (1,), generator=g))
    R, S = ex.gen_build(n), ex.gen_probe(m, n, miss_mod=int(torch.randint(0, 4, (1,), generator=g)))
    fl = flags[int(torch.randint(0, len(flags), (1,), generator=g))]
    r = ex.join_device(R, S, fl)
    assert int(r.n_matches) <= m
    if it % 500 == 0:
        free, _ = torch.cuda.mem_get_info()
        print("iter %4d: device memory in use %.1f MiB (since start %+.1f MiB), %.1f s" % (it, (total - free) / 2**20, (free0 - free) / 2**20, time.time() - t0), flush=True)
    if it % 7 == 0:
        ex.release_result()
free, _ = torch.cuda.mem_get_info()
print("end: device memory in use %.1f MiB (since start %+.1f MiB)" % ((total - free) / 2**20, (free0 - free) / 2**20))
ex.close()
torch.cuda.empty_cache()
free, _ = torch.cuda.mem_get_info()
print("after close + empty_cache: since start %+.1f MiB" % ((free0 - free) / 2**20))
