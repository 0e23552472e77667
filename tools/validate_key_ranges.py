#!/usr/bin/env python3
"""Ordered foreign-key joins at sizes where the planner cuts them into key ranges (HMJ_PATH_KEY_RANGES): the ordered result's checksums must
equal the count-mode join's (a different path: no key ranges), keys must ascend, payloads must ascend inside a key, and the time is printed.
usage: validate_key_ranges.py [log2_build log2_probe] ..."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hashmergejoin_amd as H
from hashmergejoin_amd.join import _memcpy_d2d

pairs = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(22, 30), (24, 30)]
ex = H.Executor(0)
SIGN = torch.iinfo(torch.int64).min
for kb, kp in pairs:
    R, S = ex.gen_build(1 << kb), ex.gen_uniform_domain(1 << kp, 1 << kb)
    want = ex.join_device(R, S, H.HMJ_CHECKSUM).checks()
    ms = []
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ex.join_device(R, S, H.HMJ_ORDERED | H.HMJ_CHECKSUM)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    t = ex.last_timing()
    ok = r.checks() == want
    n = int(r.n_matches)
    k = torch.empty(n, dtype=torch.int64, device="cuda")
    _memcpy_d2d(torch, k, r.key, n * 8)
    k ^= SIGN
    asc = bool((k[1:] >= k[:-1]).all())
    sv = torch.empty(n, dtype=torch.int64, device="cuda")
    _memcpy_d2d(torch, sv, r.sval, n * 8)
    sv ^= SIGN
    same = k[1:] == k[:-1]
    inner = bool((sv[1:][same] >= sv[:-1][same]).all())
    del k, sv, same
    print("2^%d x 2^%d ordered: %s ms  path %#x key_ranges=%s checksums=%s keys ascend=%s payloads ascend inside a key=%s" % (
        kb, kp, " ".join("%.1f" % x for x in ms), t["path"], bool(t["path"] & H.HMJ_PATH_KEY_RANGES), ok, asc, inner), flush=True)
    ex.release_result()
    del R, S
